#!/bin/bash
# gpurun helper (round 5): the whole -m gpu suite (RCCL and recovery tests included), then the default bench run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 3000 python -m pytest tests -m gpu -q -x --no-header --durations=15 2>&1 | tail -30 | tee gpurun_out/r5/gpu_suite.txt
if [ -z "$SKIP_BENCH" ]; then timeout 900 python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err; tail -c 1500 gpurun_out/r5/bench_default.json; fi
