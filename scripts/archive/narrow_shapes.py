#!/usr/bin/env python3
"""Step time of narrow scalers of several depths and widths, one MC sample, 4 M observations:
usage  [CARELESS_HIP_LANE=0] python scripts/narrow_shapes.py ["L,w,d ..."]   (lane-per-observation kernel against elbo_narrow.hip)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from careless_amd.workloads import build_model, reference_inputs
from careless_amd.synthetic import make_synthetic
N = 4_000_000
shapes = [tuple(int(x) for x in s.split(",")) for s in (sys.argv[1] if len(sys.argv) > 1 else "20,10,5 12,10,5 6,10,5 20,8,8 20,6,6 20,5,5 10,5,5 20,4,4").split()]
cache = {}
for L, w, d in shapes:
    if d not in cache:
        cache[d] = make_synthetic(N, d0=d, posenc=False, outliers=True)
    data = cache[d]
    model = build_model(data, L, w, 1, dof=16.0)
    eng = model.engine(reference_inputs(data)); eng.alloc_history(30)
    for i in range(5):
        eng.train_step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        eng.train_step(5 + i)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
    print("mono 4M %2dx%-2d d=%-2d S=1 studentt: %.3f ms/step %.3e refl/s LANE=%s" % (L, w, d, 1e3 * t, N / t, os.environ.get("CARELESS_HIP_LANE", "1")))
    del eng, model
