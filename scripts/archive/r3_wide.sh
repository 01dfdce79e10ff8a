#!/bin/bash
# the unfused wide path (width > 64) on its own GEMM kernels: parity, then a timed workload with a kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_wide
python -m pytest tests -m gpu -x -q -k "wide or random_engine" 2>&1 | tail -5
python bench.py --workload mono_2M_studentt_3x128_S4 --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/r3_wide/bench.err | tail -1 | tee gpurun_out/r3_wide/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_wide/prof -o wide -- python3 bench.py --workload mono_2M_studentt_3x128_S4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_wide/prof.log 2>&1
find gpurun_out/r3_wide/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_wide/kernel_stats.csv
head -12 gpurun_out/r3_wide/kernel_stats.csv
