#!/bin/bash
# kernel-trace averages of the small kernels of a step (everything but the fused scaler kernel) on three workloads
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/smallk; mkdir -p $out
for wl in ${WLS:-mono_10M_cli_default_20x10_S1 mono_1M_normal_5x64_S1 mono_10M_studentt_posenc_5x64_S8}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_$wl -o t -- python3 bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline > $out/b_$wl.json 2> $out/b_$wl.err
  f=$(find $out/p_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] || { echo "no kernel_stats.csv (the profiled command failed)"; continue 2>/dev/null || exit 1; }
  echo "== $wl: $(python3 -c "import json;d=json.loads(open('$out/b_$wl.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],4),'ms/step, kernel',round(d['roofline']['kernel_ms'],4))")"
  cut -d, -f1,2,4 $f | sed 's/"void at::native::[a-z_]*<[0-9, ]*at::native::\([A-Za-z]*\)[^"]*"/"\1"/' | cut -c1-90 | sed -n 2,9p
done
