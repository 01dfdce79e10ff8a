"""The step of a training whose scaling model is frozen, on a bench workload's data (one MI355X; round 6): event-timed step and
event-timed `cl_frozen_rows` / slot-kernel part, with the algorithmic bytes of the data term beside them.  The program rocprofv3 wraps
for profiles/r6_kernel_stats_frozen_*.csv and the PMC passes.

    python3 scripts/frozen_bench.py WORKLOAD [--steps 50] [--slot]      # --slot: round 5's cl_slot_rows path (FROZEN_SORTED_ROWS = False)
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from careless_amd.workloads import make_workload

ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--slot", action="store_true")
a = ap.parse_args()
model, inputs, data, spec = make_workload(a.workload)
model.scaling_model.trainable = False
if a.slot:
    from careless_amd.engine import ElboEngine
    ElboEngine.FROZEN_LAUE_PACKED = False              # (harmonic groups: round 5's three slot launches on plain rows)
eng = model.engine(inputs)
if a.slot:
    eng.FROZEN_SORTED_ROWS = False
eng.alloc_history(a.steps + a.warmup + 5)
for i in range(a.warmup):
    eng.train_step(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(a.warmup, a.warmup + a.steps):
    eng.train_step(i)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.steps
# the data term alone (what replaces the fused kernel): timed around _data_term of the same engine
from careless_amd import engine as E
st = E._stream()
d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
d0.record()
for i in range(a.steps):
    eng._data_term(eng.obs, i, None, None, st)
d1.record()
torch.cuda.synchronize()
dms = d0.elapsed_time(d1) / a.steps
N, S, R = int(eng.obs.N) if hasattr(eng.obs, "N") else spec["N"], eng.S, eng.R
alg = 28.0 * N + 8.0 * R * S          # rows once (refl, loc, sigma, image scale, iobs, sig, key) + z_f read and dz_f written once per (reflection, sample)
if getattr(eng.obs, "fused_laue", False) or spec.get("kind") == "laue":
    alg += (4.0 + 8.0 * S + 8.0) * N  # harmonic groups: gmeta, the per-row gradient written and read once, (refl, src) of the second pass
print(json.dumps(dict(workload=a.workload, path="cl_slot_rows" if a.slot else "cl_frozen_rows", ms_per_step=ms, data_term_ms=dms, rows=N, S=S, R=R,
                      algorithmic_GB=alg / 1e9, data_term_TBps=alg / dms / 1e9, frac_of_8TBps=alg / dms / 1e9 / 8.0, loss_finite=bool(torch.isfinite(eng.grads).all()))))
