#!/bin/bash
# round 2, first GPU pass: the whole -m gpu suite, the default bench line (with the bounded CPU baseline), the 2-rank gloo
# rehearsal of the self-launching bench on ONE GPU, and kernel-trace stats of every other BASELINE configuration
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -q -x --no-header > gpurun_out/r2a/pytest.log 2>&1; tail -3 gpurun_out/r2a/pytest.log
python bench.py > gpurun_out/r2a/bench_default.json 2> gpurun_out/r2a/bench_default.err; tail -c 1500 gpurun_out/r2a/bench_default.json
python bench.py --gpus 2 --backend gloo --nobs 2000000 --no-cpu-baseline > gpurun_out/r2a/bench_gloo2.json 2> gpurun_out/r2a/bench_gloo2.err; echo "gloo2 rc=$?"; tail -c 600 gpurun_out/r2a/bench_gloo2.json
python bench.py --gpus 1 --nobs 2000000 --no-cpu-baseline 2>/dev/null | tail -c 300
for wl in mono_1M_normal_5x64_S1 laue_5M_normal_5x64_S1 dw_50M_normal_5x64_S1 mono_10M_cli_default_20x10_S1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2a/stats_$wl -- python3 bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_$wl.json 2> gpurun_out/r2a/bench_$wl.err
  f=$(find gpurun_out/r2a/stats_$wl -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r2a/kernel_stats_$wl.csv; cut -d, -f1-4 $f | cut -c1-110 | head -4
  rm -rf gpurun_out/r2a/stats_$wl
done
