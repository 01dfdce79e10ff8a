"""Rewrite DESIGN.md section 6's table block from the stored evidence (profiles/r6_*; scripts/r6_store.sh first):

    python scripts/design_section6.py r6

Replaces everything from "All BASELINE configurations, the default scaler ..." up to "The default run without any profiler" with the table of
scripts/design_table.py in the section's row order, the source hash of profiles/TAG_sources.txt and the pass / skip counts of
profiles/TAG_gpu_suite.txt."""
import re
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
table = subprocess.run([sys.executable, "scripts/design_table.py", tag], stdout=subprocess.PIPE, text=True, check=True).stdout
lines = [l for l in table.split("\n") if l.startswith("|")]
head, rows = lines[:2], {l.split("`")[1]: l for l in lines[2:]}
ORDER = ["mono_1M_normal_5x64_S1", "mono_10M_studentt_posenc_5x64_S8", "laue_5M_normal_5x64_S1", "dw_50M_normal_5x64_S1", "mono_10M_cli_default_20x10_S1",
         "laue_5M_normal_20x10_S1", "dw_10M_normal_20x10_S1", "mono_10M_studentt_posenc_20x10_S8", "mono_10M_studentt_posenc4_20x10_S8", "mono_10M_10x10_S1",
         "mono_10M_24x10_S1", "mono_10M_20x10_img2_S1", "mono_10M_20x10_img3_S1", "mono_10M_10x10_img2_S1", "laue_5M_normal_20x10_img2_S1", "mono_10M_studentt_posenc_20x10_img2_S8",
         "mono_10M_studentt_posenc_4x64_img1_S8", "mono_2M_studentt_3x128_S4"]
NOTES = {"mono_1M_normal_5x64_S1": " (configs[1])", "mono_10M_studentt_posenc_5x64_S8": " (configs[2], **the bench line**)",
         "laue_5M_normal_5x64_S1": " (configs[3], 1 GPU, single pass)", "dw_50M_normal_5x64_S1": " (configs[4], 1 GPU)",
         "mono_10M_cli_default_20x10_S1": " (the CLI default)",
         "mono_10M_10x10_S1": " (**round 6**: the lane kernel compiled for depth 10; round 5, narrow kernel: 1.65 ms, 0.23)",
         "mono_10M_24x10_S1": " (**round 6**: a chain of two lane blocks + `cl_chain_dx`; round 5, two blocks of the 16-wide kernel: 7.72 ms)",
         "mono_10M_studentt_posenc4_20x10_S8": " (d = 37: peeled first layer)",
         "mono_10M_20x10_img3_S1": " (**round 6**: three per-image layers, a lane unit without the MFMA-in-VGPR option; the 16-wide `IMGL` instance before: 5.56 ms, 0.157)",
         "mono_10M_10x10_img2_S1": " (**round 6**: the per-image-layer instance of the depth-10 unit; the 16-wide `IMGL` instance before: 3.98 ms, 0.114)",
         "mono_10M_studentt_posenc_20x10_img2_S8": " (d = 21: peeled first layer + the dZ₀-storing per-image-layer instance, **back in round 6**; round 5: 4.43 ms, 0.206)"}
body = []
for w in ORDER + sorted(set(rows) - set(ORDER)):
    if w not in rows:
        continue
    l = rows[w]
    if w in NOTES:
        l = l.replace(f"| `{w}` |", f"| `{w}`{NOTES[w]} |", 1)
    body.append(l)
src = open(f"profiles/{tag}_sources.txt").read().strip()
suite = open(f"profiles/{tag}_gpu_suite.txt").read()
m = re.search(r"(\d+) passed, (\d+) skipped", suite)
flips = re.search(r"branch-flip resolutions this session: (\d+)", suite)
sec = f'''All BASELINE configurations, the default scaler on every data kind and round 6's new routes, one MI355X, final sources `{src}`
(`scripts/r6_final.sh` → `scripts/profiles_all.sh`: ONE `gpurun` call; GPU suite of the same call: **{m.group(1)} passed / {m.group(2)} skipped**, {flips.group(1)} branch-flip
resolution(s) — `profiles/r6_gpu_suite.txt`).  Every row: the bench line of `python3 bench.py --workload W --no-cpu-baseline` taken OUTSIDE the tracer
(`profiles/r6_bench_W.json`), the `rocprofv3 --kernel-trace --stats` summary of the same command run again (`profiles/r6_kernel_stats_W.csv`; in
brackets: the scaler's kernels summed per step), the PMC passes — separate `--pmc` runs, `profiles/r6_pmc_W.txt` → `profiles/traffic.json`.
Fractions live event-timed on the dominant kernel(s), on step time in brackets.  `profiles/INDEX.md` lists the files row by row; the table is
`python scripts/design_table.py r6`.  (Every call gets another box of the pool: the same kernel reads 0.725–0.732 on the bench line across
the round's calls.)

''' + "\n".join(head + body) + "\n\n"
s = open("DESIGN.md").read()
a = s.index("All BASELINE configurations, the default scaler on every data kind and round 6's new routes")
b = s.index("The default run without any profiler")
open("DESIGN.md", "w").write(s[:a] + sec + s[b:])
print("DESIGN.md section 6 table rewritten for sources", src)
