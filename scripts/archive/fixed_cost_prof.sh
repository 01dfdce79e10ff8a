#!/bin/bash
# kernel-trace durations (no event / dispatch overhead) of the dominant kernel on small problems: where the fixed cost of a launch goes
# (one workgroup with one tile; 64 workgroups; one tile per wave; two tiles per wave)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/fixed; mkdir -p $out
for spec in "mono_10M_cli_default_20x10_S1 128" "mono_10M_cli_default_20x10_S1 16384" "mono_10M_cli_default_20x10_S1 65536" "mono_10M_cli_default_20x10_S1 131072" \
            "mono_10M_studentt_posenc_5x64_S8 128" "mono_10M_studentt_posenc_5x64_S8 32768" "mono_10M_studentt_posenc_5x64_S8 65536"; do
  set -- $spec
  rm -rf $out/p_$1_$2
  ${PRE:-} rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_$1_$2 -o t -- python3 bench.py --workload $1 --nobs $2 --steps 50 --warmup 5 --no-cpu-baseline > $out/b.json 2> $out/b.err
  f=$(find $out/p_$1_$2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] || { echo "no kernel_stats.csv (the profiled command failed)"; continue 2>/dev/null || exit 1; }
  echo "$1 nobs=$2: $(sed -n 2,12p $f | grep -E "elbo_(lane|mlp|narrow)" | head -1 | awk -F'",' '{print $2}' | cut -d, -f1,3,5,6)  (calls, avg ns, min, max)"
done
