#!/bin/bash
# gpurun helper: step time against the sum of kernel times for small problems on the CLI-default scaler (launch gaps, small kernels)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/small; mkdir -p $O
for n in 100000 1000000; do
  python bench.py --workload mono_10M_cli_default_20x10_S1 --nobs $n --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('nobs $n: ms/step %.4f kernel ms %.4f'%(d['ms_per_step'], d['roofline']['kernel_ms']))"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$n -- python3 bench.py --workload mono_10M_cli_default_20x10_S1 --nobs $n --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
  f=$(find $O/st_$n -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | cut -c1-100 | head -9; rm -rf $O/st_$n
done
