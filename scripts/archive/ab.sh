#!/bin/bash
# A/B two builds of libcareless_hip.so on ONE device in ONE gpurun call (devices differ by several % on MFMA-bound kernels).
# usage: scripts/ab.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do
 for L in $A $B; do
  for w in mono_1M_normal_5x64_S1 mono_10M_studentt_posenc_5x64_S8; do
   CARELESS_HIP_LIB=$PWD/$L python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['config']['workload'][:8], '%.3f ms'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'])"
  done
 done
done
