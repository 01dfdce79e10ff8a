#!/bin/bash
# After scripts/r6_final_b.sh (second call of the closing evidence): the default bench run, the soak and the randomized sweep into profiles/.
set -e
cp gpurun_out/r6/bench_default_run.json profiles/r6_bench_default_run.json
src=$(cat profiles/r6_sources.txt)
{ echo "Soak on the final sources $src (scripts/r6_final_b.sh -> scripts/probe/lane_defect_probe.py --same-engine): 60 launches at 10 M observations on one engine,"
  echo "every output against the first launch's (the scaler's gradient bit for bit; amplitude gradients up to the order of their float atomics):"
  cat gpurun_out/r6/soak.txt; } > profiles/r6_soak.txt
sed -i '/^Randomized sweep/,$d' profiles/r6_gpu_suite.txt
cat gpurun_out/r6/random_sweep.txt >> profiles/r6_gpu_suite.txt
tail -c 400 profiles/r6_bench_default_run.json; echo; cat profiles/r6_soak.txt; tail -4 profiles/r6_gpu_suite.txt
