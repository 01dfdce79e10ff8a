#!/bin/bash
# gpurun helper (round 5, first call): the split-bf16 probe, then the 64-wide kernel of the round-3 tree (git worktree of e05668d
# under _r3tree/, its own library and Python) against the shipped tree on cfg2 / Laue / headline, alternating, ONE device.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
echo "== split_bf16_probe"; timeout 600 scripts/probe/split_bf16_probe 64 1024 2>&1 | tee gpurun_out/r5/split_bf16_probe.txt
line() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-40s" % sys.argv[1], "ms/step", round(d["ms_per_step"], 4), "kernel ms", round(d["roofline"].get("kernel_ms", 0), 4), "frac", round(d["roofline"]["frac"], 4), "build", d.get("build"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
if [ -d _r3tree ]; then
for rep in 1 2; do
  for WL in mono_1M_normal_5x64_S1 laue_5M_normal_5x64_S1 mono_10M_studentt_posenc_5x64_S8; do
    for tree in r3 r5; do
      if [ $tree == r3 ]; then dir=_r3tree; else dir=.; fi
      (cd $dir && timeout 900 python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline) > gpurun_out/r5/ab_${tree}_$WL.json 2> gpurun_out/r5/ab_${tree}_$WL.err || tail -3 gpurun_out/r5/ab_${tree}_$WL.err
      line "$tree $WL (round $rep)" gpurun_out/r5/ab_${tree}_$WL.json
    done
  done
done 2>&1 | tee gpurun_out/r5/mlp_r3_vs_r5_ab.txt
fi
echo "== GPU tests touched this round"; timeout 900 python -m pytest tests/test_output_step.py -q -x --no-header 2>&1 | tail -4
echo "== recovery check"
for k in mono laue dw; do
  if [ $k == mono ]; then n=1000000; else n=200000; fi
  timeout 600 python scripts/recovery_check.py $k $n 1500 0.001 0.01 0.03 2>&1 | tail -4
done | tee gpurun_out/r5/recovery_check.txt
