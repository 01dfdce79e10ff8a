#!/bin/bash
# gpurun helper (round 5): kernel trace of one peeled-first-layer shape (d = 37, 20 x 10, S = 1, 4 M observations)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
DS=${DS:-37} LS=20 WS=${WS:-10} SS=${SS:-1} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5/peel_prof -- python3 scripts/envelope.py > gpurun_out/r5/peel_prof.log 2>&1
tail -3 gpurun_out/r5/peel_prof.log
f=$(ls gpurun_out/r5/peel_prof/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-200
