#!/bin/bash
# Round 6: the evidence on the final sources in ONE call -- the -m gpu suite, bench lines + kernel-trace summaries + PMC passes of every
# workload (scripts/profiles_all.sh), the default bench run, the frozen-step table with its kernel trace and PMC traffic, the envelope
# tables, the eight-rank rehearsal.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
mkdir -p gpurun_out/r6
( time timeout 3600 python3 -m pytest tests -m gpu -q --no-header ) 2>&1 | tail -12 | tee gpurun_out/r6/gpu_suite.txt
bash scripts/profiles_all.sh 2>&1 | tail -90
timeout 1500 python3 bench.py > gpurun_out/r6/bench_default.json 2> gpurun_out/r6/bench_default.err; tail -c 900 gpurun_out/r6/bench_default.json
: > gpurun_out/r6/frozen_step.jsonl
for W in mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_5x64_S8 laue_5M_normal_5x64_S1 laue_5M_normal_20x10_S1 mono_10M_20x10_img2_S1 dw_10M_normal_20x10_S1; do
  timeout 600 python3 scripts/frozen_bench.py $W 2>/dev/null | tail -1 >> gpurun_out/r6/frozen_step.jsonl
  timeout 600 python3 scripts/frozen_bench.py $W --slot 2>/dev/null | tail -1 >> gpurun_out/r6/frozen_step.jsonl
done
cut -c1-230 gpurun_out/r6/frozen_step.jsonl
for W in mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_5x64_S8 laue_5M_normal_5x64_S1; do
  rm -rf gpurun_out/fz_$W gpurun_out/fzC_$W gpurun_out/fzD_$W
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fz_$W -- python3 scripts/frozen_bench.py $W --steps 20 > gpurun_out/r6/fz_trace.log 2>&1
  f=$(ls gpurun_out/fz_$W/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/r6/kernel_stats_frozen_$W.csv
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fzC_$W -- python3 scripts/frozen_bench.py $W --steps 4 --warmup 1 > gpurun_out/r6/fz_pmcC.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/fzD_$W -- python3 scripts/frozen_bench.py $W --steps 4 --warmup 1 > gpurun_out/r6/fz_pmcD.log 2>&1
  python3 - $W > gpurun_out/r6/pmc_frozen_$W.txt <<'PY'
import csv, glob, collections, sys
W = sys.argv[1]
for d in "CD":
    for f in glob.glob(f"gpurun_out/fz{d}_{W}/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'frozen_' in r['Kernel_Name']:
                acc[(r['Kernel_Name'].split('(')[0], r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k, c), v in sorted(acc.items()):
            print(d, k, c, "launches", len(v), "per launch", sum(v) / len(v))
print("# FETCH_SIZE / WRITE_SIZE in KiB per launch; on gfx950 FETCH_SIZE counts 64 B per 128-B request: double it (MI355X_MICROARCH.md; scripts/pmc_passes_step.sh)")
PY
  cat gpurun_out/r6/pmc_frozen_$W.txt
  rm -rf gpurun_out/fz_$W gpurun_out/fzC_$W gpurun_out/fzD_$W
done
N=4000000 LS=2,5,8,10,12,16,19,20,24,40 WS=5,6,8,10,12 DS=5,12,21 SS=1,8 timeout 2400 python3 scripts/envelope.py 2>/dev/null > gpurun_out/r6/envelope.txt
CARELESS_HIP_LANE_DEPTHS=0 CARELESS_HIP_CHAIN_LANE=0 CARELESS_HIP_LANE_BLOCKS=0 CARELESS_HIP_LANE_W12=0 N=4000000 LS=2,5,8,10,12,16,19,20,24,40 WS=5,6,8,10,12 DS=5,12,21 SS=1,8 timeout 2400 python3 scripts/envelope.py 2>/dev/null > gpurun_out/r6/envelope_before.txt
paste -d'\n' gpurun_out/r6/envelope.txt gpurun_out/r6/envelope_before.txt | grep "S=1" | cut -c1-170 | head -40
bash scripts/r6_rehearsal.sh 2>&1 | tail -12
