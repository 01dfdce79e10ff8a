#!/bin/bash
# gpurun helper (round 5): the bench line of the headline and of the peeled-first-layer workload (short runs, no CPU baseline)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5/b1.json
python bench.py --workload mono_10M_studentt_posenc4_20x10_S8 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5/b2.json
python - <<'PY'
import json
for f in ("gpurun_out/r5/b1.json", "gpurun_out/r5/b2.json"):
    d = json.loads(open(f).read()); r = d["roofline"]
    print(d["config"]["workload"], round(d["ms_per_step"], 3), r["kernel"], round(r["kernel_ms"], 3), round(r["frac"], 3), r["traffic"], r["traffic_from"])
PY
