#!/bin/bash
# round 4, seventeenth GPU pass: square-layer kernel with wave-uniform block bases + 32-bit lane offsets and operand quads half a block ahead (shipped) against the previous commit (variant library)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b17; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --no-header -x -k "wide or random_engine" 2>&1 | tail -4 ) 2>&1 | tee $O/pytest.log
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
for rep in 1 2 3; do
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_r4w_base.so timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/w_0.json 2> $O/w_0.err || tail -3 $O/w_0.err
  line "previous commit (64-bit lane pointers, quads a block ahead)" $O/w_0.json
  timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/w_1.json 2> $O/w_1.err || tail -3 $O/w_1.err
  line "uniform bases + 32-bit lane offsets, quads half a block ahead" $O/w_1.json
done 2>&1 | tee $O/wide_ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 bench.py --workload mono_2M_studentt_3x128_S4 --steps 10 --warmup 3 --no-cpu-baseline > $O/wide_bench.json 2> $O/wide_bench.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/wide_kernel_stats.csv && head -8 $O/wide_kernel_stats.csv | cut -c1-160
rm -rf $O/prof
