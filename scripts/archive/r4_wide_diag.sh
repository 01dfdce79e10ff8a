#!/bin/bash
# diagnostic (round 4): per-kernel times of the width-128 workload with the stream kernels' global loads (1), stores (2), both (3) or the
# tiled kernel's operand loads (4) compiled out (-DCL_WIDE_DIAG=n: WRONG results) -- what is left is the kernels' MFMA + LDS floor
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4wd; mkdir -p $O
for v in libcareless_hip exp_r4w_diag1 exp_r4w_diag2 exp_r4w_diag3 exp_r4w_diag4; do
  rm -rf $O/prof_$v
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o t -- python3 bench.py --workload mono_2M_studentt_3x128_S4 --steps 10 --warmup 3 --no-cpu-baseline > $O/b_$v.json 2> $O/b_$v.err
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] || { echo "$v: no stats"; tail -3 $O/b_$v.err; continue; }
  cp $f $O/stats_$v.csv; rm -rf $O/prof_$v
  python3 - $v $O/stats_$v.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[2])))
out = []
for r in rows:
    n = r["Name"]
    if "wide_" in n:
        short = n.split("::")[-1].split("(")[0]
        out.append("%s x%d %.3f" % (short, int(r["Calls"]) // 13, float(r["AverageNs"]) / 1e6))
print(sys.argv[1], " | ".join(out))
PY
done
