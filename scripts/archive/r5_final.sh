#!/bin/bash
# gpurun helper (round 5, second half): the evidence on the final sources in ONE call -- the -m gpu suite, kernel-trace summaries + bench lines
# + PMC passes of every workload (scripts/profiles_all.sh), the default bench run, the frozen-step and default-scaler tables.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 3000 python -m pytest tests -m gpu -q --no-header 2>&1 | tail -6 | tee gpurun_out/r5/gpu_suite.txt
bash scripts/profiles_all.sh 2>&1 | tail -70
timeout 1200 python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err; tail -c 700 gpurun_out/r5/bench_default.json
timeout 900 python scripts/frozen_step.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/frozen_step.txt
