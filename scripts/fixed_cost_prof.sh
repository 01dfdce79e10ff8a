#!/bin/bash
# kernel-trace durations (no event / dispatch overhead) of the dominant kernel at one and two tiles per wave: fixed cost = 2 T1 - T2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/fixed; mkdir -p $out
for spec in "mono_10M_cli_default_20x10_S1 65536" "mono_10M_cli_default_20x10_S1 131072" "mono_10M_studentt_posenc_5x64_S8 32768" "mono_10M_studentt_posenc_5x64_S8 65536"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_$1_$2 -o t -- python3 bench.py --workload $1 --nobs $2 --steps 50 --warmup 5 --no-cpu-baseline > $out/b.json 2> $out/b.err
  f=$(find $out/p_$1_$2 -name "*kernel_stats.csv" | head -1)
  echo "$1 nobs=$2: $(head -2 $f | tail -1 | cut -d, -f1-4 | cut -c1-110)"
  sed -n 2,9p $f | cut -d, -f1,2,4 | cut -c1-100
done
