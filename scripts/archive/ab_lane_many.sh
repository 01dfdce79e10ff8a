#!/bin/bash
# gpurun helper: time the CLI-default workload on several variants of the library (careless_amd/lib/exp_NAME.so)
# bash scripts/ab_lane_many.sh NAME...      (env ROUNDS, default 2; WL, default mono_10M_cli_default_20x10_S1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lane
WL=${WL:-mono_10M_cli_default_20x10_S1}
for rep in $(seq 1 ${ROUNDS:-2}); do
for v in "$@"; do
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$v.so timeout 600 python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/lane/m_$v.json 2> gpurun_out/lane/m_$v.err || tail -5 gpurun_out/lane/m_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/lane/m_$v.json").read().strip().splitlines()[-1])
    print("%-14s" % "$v", "ms/step", round(d["ms_per_step"], 3), "kernel ms", round(d["roofline"].get("kernel_ms", 0), 3), "frac", round(d["roofline"]["frac"], 4))
except Exception as e:
    print("$v failed", e)
PY
done
done
