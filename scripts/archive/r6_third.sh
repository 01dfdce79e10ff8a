#!/bin/bash
# Round 6, third GPU call: the frozen-scaler step on cl_frozen_rows -- parity tests, timing against round 5's path, kernel trace, PMC traffic.
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
mkdir -p gpurun_out
( timeout 1200 python3 -m pytest tests/test_frozen_scaler.py -m gpu -q -x ) > gpurun_out/r6_frozen_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/r6_frozen_tests.txt
tail -6 gpurun_out/r6_frozen_tests.txt
: > gpurun_out/r6_frozen_step.jsonl
for W in mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_5x64_S8 laue_5M_normal_5x64_S1 mono_10M_20x10_img2_S1 dw_10M_normal_20x10_S1; do
  timeout 600 python3 scripts/frozen_bench.py $W 2>/dev/null | tail -1 >> gpurun_out/r6_frozen_step.jsonl
  timeout 600 python3 scripts/frozen_bench.py $W --slot 2>/dev/null | tail -1 >> gpurun_out/r6_frozen_step.jsonl
done
cat gpurun_out/r6_frozen_step.jsonl | cut -c1-330
export TMPDIR=/tmp
for W in mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_5x64_S8; do
  rm -rf gpurun_out/fz_$W gpurun_out/fzC_$W gpurun_out/fzD_$W
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fz_$W -- python3 scripts/frozen_bench.py $W --steps 20 > gpurun_out/fz_trace.log 2>&1
  f=$(ls gpurun_out/fz_$W/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/r6_kernel_stats_frozen_$W.csv
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fzC_$W -- python3 scripts/frozen_bench.py $W --steps 4 --warmup 1 > gpurun_out/fz_pmcC.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/fzD_$W -- python3 scripts/frozen_bench.py $W --steps 4 --warmup 1 > gpurun_out/fz_pmcD.log 2>&1
  python3 - $W <<'PY' > gpurun_out/r6_pmc_frozen_$1.txt
import csv, glob, collections, sys
W = sys.argv[1]
for d in "CD":
    for f in glob.glob(f"gpurun_out/fz{d}_{W}/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'frozen_rows' in r['Kernel_Name'] or 'frozen_edges' in r['Kernel_Name']:
                acc[(r['Kernel_Name'].split('(')[0], r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k, c), v in sorted(acc.items()):
            print(d, k, c, "launches", len(v), "per launch", sum(v) / len(v))
print("# FETCH_SIZE / WRITE_SIZE in KiB per launch; on gfx950 FETCH_SIZE counts 64 B per 128-B request: double it (MI355X_MICROARCH.md; scripts/pmc_passes_step.sh)")
PY
  mv gpurun_out/r6_pmc_frozen_.txt gpurun_out/r6_pmc_frozen_$W.txt 2>/dev/null
  head -12 gpurun_out/r6_kernel_stats_frozen_$W.csv | cut -c1-200
  cat gpurun_out/r6_pmc_frozen_$W.txt
  rm -rf gpurun_out/fz_$W gpurun_out/fzC_$W gpurun_out/fzD_$W
done
