#!/bin/bash
# round 4: multi-rank rehearsals of bench.py on ONE GPU (gloo backend; every rank uses this device): 2 ranks (row split), 4 ranks (reflection-owner
# split) with the extra configuration of 4 ranks (Laue) at a small size, 2 ranks with the two-piece message
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4reh; mkdir -p $O
run() {  # tag, env, args...
  tag=$1; shift; envs=$1; shift
  env $envs timeout 900 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err; rc=$?
  python3 - $tag $rc $O/$tag.json <<'PY'
import json, sys
tag, rc, f = sys.argv[1:4]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ex = {k: (v.get("value"), v.get("config", {}).get("parallelism")) for k, v in d.get("extra_configs", {}).items()}
    print("%-28s rc=%s ranks %s value %.3e ms/step %.4f %s obs/rank %s loss_finite %s extras %s" % (tag, rc, d["ranks_seen"], d["value"] or 0, d["ms_per_step"], d["config"]["parallelism"], d["obs_per_rank"], d["config"]["loss_finite"], ex))
except Exception as e:
    print(tag, "rc=%s" % rc, "failed", e)
PY
  [ $rc -ne 0 ] && tail -5 $O/$tag.err
}
run gloo2_rows X=0 --gpus 2 --backend gloo --nobs 2000000 --steps 5 --warmup 2 --no-cpu-baseline
run gloo2_two_piece CARELESS_HIP_SPLIT_MESSAGE=1 --gpus 2 --backend gloo --nobs 2000000 --steps 5 --warmup 2 --no-cpu-baseline
run gloo4_owner_laue_extra X=0 --gpus 4 --backend gloo --nobs 2000000 --steps 5 --warmup 2 --no-cpu-baseline --extra laue_5M_normal_5x64_S1 --extra-nobs 400000
run gloo2_default_scaler X=0 --gpus 2 --backend gloo --workload mono_10M_cli_default_20x10_S1 --nobs 1000000 --steps 5 --warmup 2 --no-cpu-baseline
