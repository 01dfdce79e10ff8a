#!/bin/bash
# gpurun helper (round 5): the split-bf16 probe (all variants) and the recovery check
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
echo "== split_bf16_probe"; timeout 600 scripts/probe/split_bf16_probe 64 256 2>&1 | tee gpurun_out/r5/split_bf16_probe.txt
if [ -z "$SKIP_RECOVERY" ]; then
echo "== recovery check"
for k in mono laue dw; do
  if [ $k == mono ]; then n=1000000; else n=200000; fi
  timeout 900 python scripts/recovery_check.py $k $n 1500 0.001 0.01 0.03 2>&1 | tail -4
done | tee gpurun_out/r5/recovery_check.txt
fi
