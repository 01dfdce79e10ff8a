#!/bin/bash
# rocprofv3 evidence for the bench line: kernel-trace stats of the default bench command, then the PMC passes (separate runs).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r1b}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_under_rocprof.json 2>gpurun_out/${TAG}_bench_err.log
f=$(find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${TAG}_kernel_stats.csv; cut -d, -f1-4 $f | cut -c1-120 | head -8
tail -1 gpurun_out/${TAG}_bench_under_rocprof.json | cut -c1-300
bash scripts/pmc_passes.sh > gpurun_out/${TAG}_pmc.txt 2>&1; tail -25 gpurun_out/${TAG}_pmc.txt
