#!/bin/bash
# round 4, tenth GPU pass: wide path, LDS operand layouts without bank conflicts under the hardware's ds_read_b128 lane groups
# (CL_WIDE_SWZ: 0 = round-3 padded pitches, 1 = streaming kernels' weight image, 2 = tiled kernel's tiles, 3 = both = shipped)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b10; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --no-header -x -k "wide or random_engine or image_layers" 2>&1 | tail -6 ) 2>&1 | tee $O/pytest.log
line() {
python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-60s ms/step %.4f kernel ms %.4f frac %.4f step frac %.4f" % (sys.argv[1], d["ms_per_step"], r["kernel_ms"], r["frac"], r["frac_on_step_time"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2; do
  for v in 0 1 2; do
    CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_r4w_swz$v.so timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/w_$v.json 2> $O/w_$v.err || tail -3 $O/w_$v.err
    line "CL_WIDE_SWZ=$v" $O/w_$v.json
  done
  timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/w_3.json 2> $O/w_3.err || tail -3 $O/w_3.err
  line "CL_WIDE_SWZ=3 (shipped)" $O/w_3.json
done 2>&1 | tee $O/wide_ab.log
for v in 0 3; do
  L=$PWD/careless_amd/lib/exp_r4w_swz$v.so; [ $v = 3 ] && L=$PWD/careless_amd/lib/libcareless_hip.so
  export CARELESS_HIP_LIB=$L
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o t -- python3 bench.py --workload mono_2M_studentt_3x128_S4 --steps 10 --warmup 3 --no-cpu-baseline > $O/wide_bench_$v.json 2> $O/wide_bench_$v.err
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/wide_kernel_stats_swz$v.csv && head -8 $O/wide_kernel_stats_swz$v.csv | cut -c1-150
  rm -rf $O/prof_$v
done
unset CARELESS_HIP_LIB
