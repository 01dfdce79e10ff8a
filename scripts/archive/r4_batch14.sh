#!/bin/bash
# round 4, fourteenth GPU pass: 64-wide fused kernel, the two accumulators of a block pair as two back-to-back chains of four MFMAs instead of alternating step by step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b14; mkdir -p $O
V=${V:-r4_chain1}
line() {
python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-44s ms/step %.4f kernel ms %.4f frac %.4f step frac %.4f  %s" % (sys.argv[1], d["ms_per_step"], r["kernel_ms"], r["frac"], r["frac_on_step_time"], r.get("kernel", "")[:60]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for wl in mono_10M_studentt_posenc_5x64_S8 laue_5M_normal_5x64_S1; do
for rep in 1 2; do
  timeout 900 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > $O/a.json 2> $O/a.err || tail -3 $O/a.err
  line "shipped $wl" $O/a.json
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$V.so timeout 900 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > $O/b.json 2> $O/b.err || tail -3 $O/b.err
  line "$V $wl" $O/b.json
done
done 2>&1 | tee $O/ab.log
( CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$V.so timeout 1500 python -m pytest tests -m gpu -q --no-header -x -k "5x64 or golden or mlp5 or studentt" 2>&1 | tail -4 ) 2>&1 | tee $O/pytest.log
