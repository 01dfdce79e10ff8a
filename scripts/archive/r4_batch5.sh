#!/bin/bash
# round 4, fifth GPU pass: the wide path with the square-layer streaming kernel and the transposed staging of the weight gradient
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b5; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --no-header -x -k "wide or random_engine or golden or trajectory or output" 2>&1 | tail -12 ) > $O/pytest.log 2>&1
cat $O/pytest.log
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
for rep in 1 2; do
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_r4w_old.so timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/w_old.json 2> $O/w_old.err || tail -3 $O/w_old.err
  line "old wide_gemm.hip (round-4 commit fa08310)" $O/w_old.json
  CARELESS_HIP_WIDE_SQ=0 timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/w_tr.json 2> $O/w_tr.err || tail -3 $O/w_tr.err
  line "register-transposed swizzled wgrad staging only (WIDE_SQ=0)" $O/w_tr.json
  timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/w_new.json 2> $O/w_new.err || tail -3 $O/w_new.err
  line "+ square-layer streaming kernel (shipped)" $O/w_new.json
done 2>&1 | tee $O/wide_ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_wide -o t -- python3 bench.py --workload mono_2M_studentt_3x128_S4 --steps 10 --warmup 3 --no-cpu-baseline > $O/wide_bench.json 2> $O/wide_bench.err
f=$(find $O/prof_wide -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/wide_kernel_stats.csv && head -12 $O/wide_kernel_stats.csv | cut -c1-150
rm -rf $O/prof_wide
