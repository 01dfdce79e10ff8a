#!/bin/bash
# gpurun helper (round 5): the flush pointers re-read from the kernel-argument block (new) against the library before (exp_prev.so) and
# the round-3 tree, alternating on one device; then the envelope table of the default scaler's neighbourhood.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
line() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-46s" % sys.argv[1], "ms/step", round(d["ms_per_step"], 4), "kernel ms", round(d["roofline"].get("kernel_ms", 0), 4), "frac", round(d["roofline"]["frac"], 4), "build", d.get("build"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
if [ -z "$SKIP_AB" ]; then
for rep in 1 2 3; do
  for WL in laue_5M_normal_5x64_S1 mono_10M_studentt_posenc_5x64_S8; do
    for tree in r3 prev new; do
      if [ $tree == r3 ]; then dir=_r3tree; else dir=.; fi
      if [ $tree == prev ]; then export CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_prev.so; else unset CARELESS_HIP_LIB; fi
      [ -d $dir ] || continue
      (cd $dir && timeout 900 python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline) > gpurun_out/r5/ab2_${tree}_$WL.json 2> gpurun_out/r5/ab2_${tree}_$WL.err || tail -3 gpurun_out/r5/ab2_${tree}_$WL.err
      line "$tree $WL (round $rep)" gpurun_out/r5/ab2_${tree}_$WL.json
    done
  done
done 2>&1 | tee gpurun_out/r5/mlp_flush_args_ab.txt
unset CARELESS_HIP_LIB
fi
if [ -z "$SKIP_ENV" ]; then timeout 3000 python scripts/envelope.py 2>&1 | tee gpurun_out/r5/envelope.txt; fi
