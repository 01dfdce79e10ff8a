"""Soak of the layer-by-layer path (round 4): 300 Adam steps of a 3 x 128 scaler on 20 000 observations, twice in deterministic mode (bit-identical
parameters and histories expected) and once in the default mode (same loss curve to the atomics' rounding); prints the verdicts."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from careless_amd.engine import ElboEngine

kw = dict(N=20000, R=900, d0=5, L=3, w=128, S=4, likelihood="studentt", dof=16.0, n_images=20)
data, cfg, params, x, u_f, eta = util.make_problem(**kw)
inputs = util.reference_inputs(data)
runs = []
for det in (True, True, False):
    m = util.build_model(data, cfg, params, 3, 128)
    m.deterministic = det
    e = ElboEngine(m, inputs, seed=3)
    e.alloc_history(300)
    for i in range(300):
        e.train_step(i)
    torch.cuda.synchronize()
    h = e.read_history(300)
    runs.append((e.params.clone(), np.asarray(h["loss"]), np.asarray(h["NLL"])))
(p0, l0, n0), (p1, l1, n1), (p2, l2, n2) = runs
print("deterministic runs bit-identical:", bool(torch.equal(p0, p1)) and bool((l0 == l1).all()))
print("loss first / last:", float(l0[0]), float(l0[-1]), "finite:", bool(np.isfinite(l0).all() and np.isfinite(l2).all()))
print("default mode vs deterministic, max relative loss difference over 300 steps: %.2e" % float(np.max(np.abs(l2 - l0) / np.abs(l0))))
ok = bool(torch.equal(p0, p1)) and np.isfinite(l0).all() and l0[-1] < l0[0] and np.max(np.abs(l2 - l0) / np.abs(l0)) < 1e-3
print("SOAK", "ok" if ok else "FAILED")
sys.exit(0 if ok else 1)
