#!/bin/bash
# gpurun helper (round 5): the whole -m gpu suite with the peeled first layer in front of both default-scaler kernels, then the envelope table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 3000 python -m pytest tests -m gpu -q -x --no-header 2>&1 | tail -12 | tee gpurun_out/r5/gpu_suite.txt
if [ -z "$SKIP_ENV" ]; then timeout 3000 python scripts/envelope.py 2>&1 | tee gpurun_out/r5/envelope.txt | tail -5; fi
