#!/usr/bin/env python3
"""Step time of the headline workload when the rows arrive in another order than the synthetic generator's (images sorted,
reflections random): sorted by reflection (an HKL-sorted unmerged file) or fully shuffled.
Usage: python scripts/sorted_probe.py [n_obs]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from careless_amd.workloads import make_workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
for order in ("as generated", "sorted by reflection", "shuffled"):
    model, inputs, data, spec = make_workload("mono_10M_studentt_posenc_5x64_S8", N=n)
    if order != "as generated":
        rid = np.asarray(inputs[0]).reshape(-1)
        perm = np.argsort(rid, kind="stable") if order == "sorted by reflection" else np.random.default_rng(0).permutation(len(rid))
        inputs = tuple(np.ascontiguousarray(np.asarray(c)[perm]) for c in inputs)
    eng = model.engine(inputs)
    eng.alloc_history(40)
    for i in range(5):
        eng.train_step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5, 25):
        eng.train_step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{order:22s} {dt * 1e3:8.3f} ms/step  {n / dt:.3e} refl/s")
    model._engine = None
    del eng
