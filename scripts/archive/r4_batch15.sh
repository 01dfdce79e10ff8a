#!/bin/bash
# round 4, fifteenth GPU pass: wide path, forward instance of the square-layer kernel with the output blocks in the outer loop (variant library) against the shipped order
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b15; mkdir -p $O
V=${V:-r4w_aout}
line() {
python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-44s ms/step %.4f kernel ms %.4f frac %.4f step frac %.4f" % (sys.argv[1], d["ms_per_step"], r["kernel_ms"], r["frac"], r["frac_on_step_time"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do
  timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/a.json 2> $O/a.err || tail -3 $O/a.err
  line "shipped" $O/a.json
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$V.so timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/b.json 2> $O/b.err || tail -3 $O/b.err
  line "$V" $O/b.json
done 2>&1 | tee $O/ab.log
for v in shipped $V; do
  L=$PWD/careless_amd/lib/exp_$v.so; [ $v = shipped ] && L=$PWD/careless_amd/lib/libcareless_hip.so
  export CARELESS_HIP_LIB=$L
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o t -- python3 bench.py --workload mono_2M_studentt_3x128_S4 --steps 10 --warmup 3 --no-cpu-baseline > $O/wb_$v.json 2> $O/wb_$v.err
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$v.csv && echo "== $v" && head -8 $O/kernel_stats_$v.csv | cut -c1-160
  rm -rf $O/prof_$v
done
( CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$V.so timeout 1500 python -m pytest tests -m gpu -q --no-header -x -k "wide" 2>&1 | tail -4 ) 2>&1 | tee $O/pytest.log
