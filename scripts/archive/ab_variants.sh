#!/bin/bash
# A/B of experimental libraries on chosen workloads, one device, two rounds: bash scripts/ab_variants.sh "wl1 wl2" variantA variantB ...
wls=$1; shift
for r in 1 2; do for v in "$@"; do for w in $wls; do
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$v.so python3 bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('exp_$v', d['config']['workload'], '%.4f ms'%d['ms_per_step'], 'kernel %.4f'%d['roofline']['kernel_ms'], 'frac %.4f'%d['roofline']['frac'])"
done; done; done
