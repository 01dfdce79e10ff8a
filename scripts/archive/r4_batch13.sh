#!/bin/bash
# round 4, thirteenth GPU pass: wide path, predict + likelihood + gradient of rows that are their own slot in one launch (CARELESS_HIP_SLOT_ROWS=0: three launches)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b13; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --no-header -x -k "wide or random_engine or image_layers or validation or output" 2>&1 | tail -6 ) 2>&1 | tee $O/pytest.log
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
  CARELESS_HIP_SLOT_ROWS=0 timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/w_0.json 2> $O/w_0.err || tail -3 $O/w_0.err
  line "predict / likelihood / backward launches (SLOT_ROWS=0)" $O/w_0.json
  timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/w_1.json 2> $O/w_1.err || tail -3 $O/w_1.err
  line "one launch for rows that are their own slot" $O/w_1.json
done 2>&1 | tee $O/wide_ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 bench.py --workload mono_2M_studentt_3x128_S4 --steps 10 --warmup 3 --no-cpu-baseline > $O/wide_bench.json 2> $O/wide_bench.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/wide_kernel_stats.csv && head -12 $O/wide_kernel_stats.csv | cut -c1-150
rm -rf $O/prof
