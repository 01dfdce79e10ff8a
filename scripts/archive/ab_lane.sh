#!/bin/bash
# gpurun helper: parity of the narrow-scaler cases on the lane-per-observation kernel, then A/B against elbo_narrow.hip
# (CARELESS_HIP_LANE=0) on the CLI-default workload.  bash scripts/ab_lane.sh [pytest -k expression]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lane
K=${1:-"cli_default or narrow or mlp9x7 or mlp7x12 or mlp5x13 or trajectory or three_obs or rank_shards or laue or ev11"}
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x --no-header -k "$K" 2>&1 | tail -15
for rep in 1 2; do
for v in 1 0; do
  CARELESS_HIP_LANE=$v timeout 600 python bench.py --workload mono_10M_cli_default_20x10_S1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/lane/ab_$v.json 2> gpurun_out/lane/ab_$v.err || tail -5 gpurun_out/lane/ab_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/lane/ab_$v.json").read().strip().splitlines()[-1])
    print("LANE=$v", d["roofline"].get("kernel"), "ms/step", round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 4), "value", "%.3e" % d["value"])
except Exception as e:
    print("LANE=$v failed", e)
PY
done
done
