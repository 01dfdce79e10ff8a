#!/usr/bin/env python3
"""Step time of a Laue problem on the careless CLI's default scaler (20 x 10): usage  [CARELESS_HIP_NARROW=0] python scripts/laue_default_scaler.py [rows]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from careless_amd.synthetic import make_synthetic_laue
from careless_amd.workloads import build_model, reference_inputs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
data = make_synthetic_laue(N, seed=1234)
model = build_model(data, 20, 10, 1, kind="laue")
col = lambda a, t: np.asarray(a).astype(t)[:, None]
inputs = reference_inputs(data) + (col(data["wavelength"], np.float32), col(data["harmonic_id"], np.int64))
eng = model.engine(inputs)
eng.alloc_history(30)
for i in range(5):
    eng.train_step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20):
    eng.train_step(5 + i)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
print("laue %d rows, 20x10 scaler: %.3f ms/step, %.3e rows/s, single pass %s, CARELESS_HIP_LANE=%s CARELESS_HIP_NARROW=%s" % (N, 1e3 * t, N / t, eng.obs.fused_laue, os.environ.get("CARELESS_HIP_LANE", "1"), os.environ.get("CARELESS_HIP_NARROW", "1")))
