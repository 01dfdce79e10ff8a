#!/bin/bash
# gpurun helper (round 5): the round's evidence on the final sources in ONE call -- the whole -m gpu suite, kernel-trace summaries + bench
# lines + PMC passes of every workload (scripts/profiles_all.sh), the default bench run, the envelope tables.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 3000 python -m pytest tests -m gpu -q --no-header 2>&1 | tail -6 | tee gpurun_out/r5/gpu_suite.txt
bash scripts/profiles_all.sh 2>&1 | tail -60
timeout 1200 python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err; tail -c 600 gpurun_out/r5/bench_default.json
timeout 3000 python scripts/envelope.py > gpurun_out/r5/envelope.txt 2>&1; tail -3 gpurun_out/r5/envelope.txt
LS=5,10,20,24 DS=5,21 WS=16,20,32,48,64 SS=1 timeout 1500 python scripts/envelope.py > gpurun_out/r5/envelope_wide.txt 2>&1
