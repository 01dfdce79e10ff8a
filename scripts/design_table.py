"""DESIGN.md section 6's table and profiles/INDEX.md from the round's evidence files (scripts/profiles_all.sh -> scripts/store_profiles.sh TAG):

    python scripts/design_table.py r6            # prints the table rows; writes profiles/INDEX.md

Per workload: the bench line taken OUTSIDE the tracer (profiles/TAG_bench_W.json), the rocprofv3 kernel-trace summary of the same command
(profiles/TAG_kernel_stats_W.csv: per-step sum over the scaler's kernels), the PMC traffic (profiles/traffic.json)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
P = "profiles"
KERNELS = ("elbo_mlp", "elbo_narrow", "elbo_lane", "wide_", "peel_", "chain_dx")
traffic = json.load(open(f"{P}/traffic.json"))
src = open(f"{P}/{tag}_sources.txt").read().strip() if os.path.exists(f"{P}/{tag}_sources.txt") else "?"
rows, index = [], []
for f in sorted(os.listdir(P)):
    if not (f.startswith(f"{tag}_bench_") and f.endswith(".json")) or "default_run" in f:
        continue
    wl = f[len(tag) + 7:-5]
    try:
        d = json.loads(open(f"{P}/{f}").read().strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001
        print("skip", f, e)
        continue
    r = d["roofline"]
    steps_traced = 13
    per_step = None
    ks = f"{P}/{tag}_kernel_stats_{wl}.csv"
    top = ""
    if os.path.exists(ks):
        tot = 0.0
        for row in csv.DictReader(open(ks)):
            if any(k in row["Name"] for k in KERNELS):
                tot += float(row["TotalDurationNs"])
                if not top:
                    top = row["Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
                    steps_traced = int(row["Calls"]) if "elbo_" in row["Name"] else steps_traced
        per_step = tot / steps_traced / 1e6
    alg = None
    try:
        alg = r["hbm_secondary"]["bytes_per_obs"] * r["obs_per_launch"] / 1e9        # 4 (d + 3) + 4 + 8 S bytes per observation
    except Exception:      # noqa: BLE001
        pass
    t = traffic.get(wl)
    tr = f"{t['hbm_bytes_per_launch'] / 1e9:.2f}" if t else "–"
    rows.append((wl, d["value"], d["ms_per_step"], r.get("kernel_ms"), per_step, r["frac"], r.get("frac_on_step_time"), tr, alg, r.get("kernel", top)))
    index.append((wl, f, os.path.basename(ks) if os.path.exists(ks) else "–", f"{tag}_pmc_{wl}.txt" if os.path.exists(f"{P}/{tag}_pmc_{wl}.txt") else "–"))
print("| workload (`bench.py --workload`) | refl/s | ms/step | dominant-kernel ms, live (rocprofv3 per step) | MFMA frac (on step time) | HBM traffic / algorithmic GB | kernel |")
print("|---|---|---|---|---|---|---|")
for wl, v, ms, kms, ps, fr, frs, tr, alg, kn in rows:
    print(f"| `{wl}` | {v:.3g} | {ms:.3f} | {kms:.3f} ({ps:.3f}) | {fr:.3f} ({frs:.3f}) | {tr} / {alg:.2f} | `{str(kn)[:90]}` |" if ps is not None and alg is not None else
          f"| `{wl}` | {v:.3g} | {ms:.3f} | {kms} | {fr:.3f} | {tr} | `{str(kn)[:90]}` |")
with open(f"{P}/INDEX.md", "w") as out:
    out.write(f"# profiles/ — which file backs which number (current round: {tag}, library sources `{src}`)\n\n")
    out.write("Files of earlier rounds (`r1_` … `r5_`) stay as history; DESIGN.md quotes them only where it says so.  Everything below was written by ONE\n"
              f"`gpurun` call on the sources above (`scripts/{tag}_final.sh` → `scripts/profiles_all.sh`, stored with `scripts/store_profiles.sh {tag}`).\n\n")
    out.write("## DESIGN §6 table: one row per workload\n\n| workload | bench line (outside the tracer) | rocprofv3 kernel-trace summary | PMC passes |\n|---|---|---|---|\n")
    for wl, b, k, pmc in index:
        out.write(f"| `{wl}` | `{b}` | `{k}` | `{pmc}` |\n")
    out.write(f"\n`traffic.json` = HBM bytes per step of every workload from the PMC files above (2 × FETCH_SIZE + WRITE_SIZE), with the source hash;\n"
              f"`{tag}_profiles_summary.txt` = the call's own one-line-per-workload summary; `{tag}_sources.txt` = the hash.\n\n")
    extra = [
        (f"{tag}_bench_default_run.json", "the default `python bench.py` run (with the CPU baseline; `traffic_from.stale` false) — DESIGN §6"),
        (f"{tag}_gpu_suite.txt", "`pytest -m gpu` of the same call, with the count of LeakyReLU branch-flip resolutions; below it the randomized sweeps of the second call (600 draws on two more seeds)"),
        (f"{tag}_lane_defect.txt", "the run-to-run defect: hazard probe on the hardware + A/B of the withdrawn instance's variants — DESIGN §4.14, NOTEBOOK R6.1"),
        (f"{tag}_ab_lrelu.txt", "round 5's library against round 6's (LeakyReLU as a compiler-known v_max_f32), alternating — DESIGN §4.14"),
        (f"{tag}_lane_repeat.txt", "tests/test_lane_repeat.py on the hardware (first run of the round)"),
        (f"{tag}_lane_repeat_8engines.txt", "the 4 M-row half of tests/test_lane_repeat.py with EIGHT fresh engines for every one of the 100 instance kinds (the suite does that for a dozen, two engines x four launches for all)"),
        (f"{tag}_soak.txt", "60 launches at 10 M observations of eight kernel instances (CLI default, dZ0-storing, per-image layers + dZ0, another depth, a lane-block chain, per-image layers at another depth, the headline kernel, 12 x 12) against the first; second call on the same sources (scripts/r6_final_b.sh)"),
        (f"{tag}_frozen_step.txt", "the frozen-scaler step, `cl_frozen_rows` against round 5's slot kernels — DESIGN §5.1b"),
        (f"{tag}_kernel_stats_frozen_*.csv / {tag}_pmc_frozen_*.txt", "kernel trace and PMC traffic of the frozen step"),
        (f"{tag}_envelope.txt / {tag}_envelope_before.txt", "depth × width × columns × samples around the default scaler, with the per-depth lane units and lane-block chains on / off — DESIGN §4.4d, §4.6"),
        (f"{tag}_imgl_depth.txt", "`--mlp-layers D --image-layers 2` on the per-depth units' per-image-layer instances, on / off (first run on the hardware) — DESIGN §4.4c"),
        (f"{tag}_imgl3.txt", "three per-image layers on the lane kernel (default depth): cases + `--image-layers 3` at 10 M observations with the lane kernel on / off — DESIGN §4.4c"),
        (f"{tag}_imgl_abort_probe.txt", "the GPU memory fault of `elbo_mlp_kernel<16, 64, 24, 0, image layers>` (a random draw on a new seed): which variations fault — DESIGN §4.7"),
        (f"{tag}_imgl_det.txt", "deterministic mode with per-image layers on the lane kernel's instances: cases + what the mode costs — DESIGN §4.10"),
        (f"{tag}_e2e.txt", "`--merge-half-datasets` and `--mlp-layers 10` through the command line on 5 M observations, wall time by stage — DESIGN §5.1b"),
        (f"{tag}_rehearsal_gloo.txt", "eight-rank gloo rehearsal of `bench.py --gpus N` + rank 0's shard of 8-rank jobs on one device — DESIGN §5.2"),
    ]
    out.write("## Other current files\n\n| file | what |\n|---|---|\n")
    for f, what in extra:
        out.write(f"| `{f}` | {what} |\n")
print(f"wrote {P}/INDEX.md")
