#!/bin/bash
# gpurun helper (round 5): gloo rehearsals of `bench.py --gpus N` on the one-GPU box (ranks share the device): 2 and 4 ranks, row split
# (the default) and reflection owners, with the extra configuration of 4 ranks at a rehearsal size; the loss histories against one rank's
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
A="--nobs 2000000 --steps 5 --warmup 1 --no-cpu-baseline"
timeout 900 python bench.py $A > gpurun_out/r5/reh_1.json 2> gpurun_out/r5/reh_1.err
timeout 900 python bench.py --gpus 2 --backend gloo $A --extra none > gpurun_out/r5/reh_2rows.json 2> gpurun_out/r5/reh_2rows.err
CARELESS_HIP_OWNER_SHARD=1 timeout 900 python bench.py --gpus 2 --backend gloo $A --extra none > gpurun_out/r5/reh_2own.json 2> gpurun_out/r5/reh_2own.err
timeout 1500 python bench.py --gpus 4 --backend gloo $A --extra laue_5M_normal_5x64_S1 --extra-nobs 400000 > gpurun_out/r5/reh_4rows.json 2> gpurun_out/r5/reh_4rows.err
python - <<'PY' | tee gpurun_out/r5/rehearsal.txt
import json, numpy as np
one = json.loads(open("gpurun_out/r5/reh_1.json").read().strip().splitlines()[-1])
for f in ("reh_2rows", "reh_2own", "reh_4rows"):
    try:
        d = json.loads(open(f"gpurun_out/r5/{f}.json").read().strip().splitlines()[-1])
        dev = float(np.max(np.abs(np.array(d["loss_history"]) / np.array(one["loss_history"]) - 1)))
        print(f, "ranks_seen", d["ranks_seen"], "backend", d["backend"], d["config"]["parallelism"], "obs_per_rank", d["obs_per_rank"], "ms/step %.3f" % d["ms_per_step"],
              "max |loss / one-rank loss - 1| %.1e" % dev, "extra", {k: (round(v.get("ms_per_step", 0), 3) if "ms_per_step" in v else v) for k, v in d.get("extra_configs", {}).items()})
    except Exception as e:
        print(f, "FAILED", repr(e)); print(open(f"gpurun_out/r5/{f}.err").read()[-600:])
PY
