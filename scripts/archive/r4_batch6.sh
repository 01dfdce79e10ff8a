#!/bin/bash
# round 4, sixth GPU pass: GPU suite after the switch clean-up of elbo_mlp.hip; bench line (must not move); tiled kernel with one LDS copy
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b6; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -8 ) > $O/pytest.log 2>&1
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
  for WL in mono_10M_studentt_posenc_5x64_S8 mono_1M_normal_5x64_S1; do
    timeout 600 python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline > $O/h.json 2> $O/h.err || tail -3 $O/h.err
    line "$WL" $O/h.json
  done
  timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/w2.json 2> $O/w2.err || tail -3 $O/w2.err
  line "wide, two LDS copies (shipped)" $O/w2.json
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_r4w_nbuf1.so timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/w1.json 2> $O/w1.err || tail -3 $O/w1.err
  line "wide, one LDS copy (CL_WIDE_NBUF=1)" $O/w1.json
done 2>&1 | tee $O/ab.log
