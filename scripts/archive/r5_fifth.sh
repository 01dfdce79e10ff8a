#!/bin/bash
# gpurun helper (round 5): the peeled first layer -- parity, then the envelope rows it touches
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 1800 python -m pytest tests/test_gpu_parity.py -q -x --no-header -k "peel or lane or cli_default" 2>&1 | tail -15
DS=${DS:-37,53} LS=${LS:-20} WS=${WS:-4,8,10} timeout 1200 python scripts/envelope.py 2>&1 | tee gpurun_out/r5/envelope_peel.txt
