#!/bin/bash
# Round 6: the multi-rank bench path rehearsed at EIGHT ranks before any node has run it (gloo backend, the ranks share the one MI355X of the
# box; reduced sizes): `bench.py --gpus 8` in both splits with the extra configuration of 8 ranks (double-Wilson, BASELINE configs[4]),
# `--gpus 4` with the extra configuration of 4 ranks (Laue, configs[3]); every loss history against the one-rank run of the same workload
# and size.  Then rank 0's shard of an 8-rank configs[4] job at FULL size on this one device (--sim-world 8: the replicated tn_* / Adam
# launches over R = 1.56 M reflections beside 1/8 of the observations), next to the headline's.
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
O=gpurun_out/r6reh; mkdir -p $O
A="--nobs 2000000 --steps 5 --warmup 1 --no-cpu-baseline"
timeout 900 python3 bench.py $A > $O/one_head.json 2> $O/one_head.err
timeout 900 python3 bench.py --workload laue_5M_normal_5x64_S1 --nobs 400000 --steps 5 --warmup 1 --no-cpu-baseline > $O/one_laue.json 2> $O/one_laue.err
timeout 900 python3 bench.py --workload dw_50M_normal_5x64_S1 --nobs 800000 --steps 5 --warmup 1 --no-cpu-baseline > $O/one_dw.json 2> $O/one_dw.err
timeout 1800 python3 bench.py --gpus 8 --backend gloo $A --extra dw_50M_normal_5x64_S1 --extra-nobs 800000 > $O/r8_rows.json 2> $O/r8_rows.err
CARELESS_HIP_OWNER_SHARD=1 timeout 1800 python3 bench.py --gpus 8 --backend gloo $A --extra none > $O/r8_own.json 2> $O/r8_own.err
timeout 1800 python3 bench.py --gpus 4 --backend gloo $A --extra laue_5M_normal_5x64_S1 --extra-nobs 400000 > $O/r4_rows.json 2> $O/r4_rows.err
timeout 1800 python3 bench.py --workload dw_50M_normal_5x64_S1 --sim-world 8 --force-dist --steps 20 --warmup 3 --no-cpu-baseline > $O/sim8_dw.json 2> $O/sim8_dw.err
timeout 1800 python3 bench.py --sim-world 8 --force-dist --steps 20 --warmup 3 --no-cpu-baseline > $O/sim8_head.json 2> $O/sim8_head.err
python3 - <<'PY' | tee gpurun_out/r6_rehearsal_gloo.txt
import json, numpy as np
O = "gpurun_out/r6reh/"
def last(f):
    return json.loads(open(O + f).read().strip().splitlines()[-1])
print("# bash scripts/r6_rehearsal.sh (one MI355X shared by the ranks, gloo backend; headline workload at 2 M observations, configs[3] at 0.4 M, configs[4] at 0.8 M; 5 steps)")
one = {"head": last("one_head.json"), "laue_5M_normal_5x64_S1": last("one_laue.json"), "dw_50M_normal_5x64_S1": last("one_dw.json")}
for f in ("r8_rows", "r8_own", "r4_rows"):
    try:
        d = last(f + ".json")
        dev = float(np.max(np.abs(np.array(d["loss_history"]) / np.array(one["head"]["loss_history"]) - 1)))
        line = [f, "ranks_seen", d["ranks_seen"], "backend", d["backend"], d["config"]["parallelism"], "obs_per_rank", d["obs_per_rank"], "ms/step %.3f" % d["ms_per_step"],
                "max |loss / one-rank loss - 1| %.1e" % dev]
        for k, v in d.get("extra_configs", {}).items():
            if "loss_history" in v:
                n = min(len(v["loss_history"]), len(one[k]["loss_history"]))
                dv = float(np.max(np.abs(np.array(v["loss_history"][:n]) / np.array(one[k]["loss_history"][:n]) - 1)))
                line += ["| extra", k, v["config"]["parallelism"], "obs_per_rank", v["obs_per_rank"], "ms/step %.3f" % v["ms_per_step"], "max |loss / one-rank loss - 1| %.1e" % dv]
            else:
                line += ["| extra", k, str(v)[:200]]
        print(*line)
    except Exception as e:
        print(f, "FAILED", repr(e)); print(open(O + f + ".err").read()[-800:])
for f, what in (("sim8_dw", "rank 0's shard of an 8-rank configs[4] job (50 M observations, R = 1.56 M), one device"), ("sim8_head", "rank 0's shard of an 8-rank headline job")):
    try:
        d = last(f + ".json")
        print(f, what, "| obs on this rank", d["obs_per_rank"], "ms/step %.3f" % d["ms_per_step"], "kernel ms", d["roofline"].get("kernel_ms"), d.get("diagnostic", ""))
    except Exception as e:
        print(f, "FAILED", repr(e)); print(open(O + f + ".err").read()[-800:])
PY
