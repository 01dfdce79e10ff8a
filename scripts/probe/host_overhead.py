import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from careless_amd.synthetic import make_synthetic
from careless_amd.workloads import build_model, reference_inputs
for N in (50_000, 200_000, 1_000_000):
    data = make_synthetic(N, d0=5, posenc=False, outliers=False)
    inputs = reference_inputs(data)
    model = build_model(data, 20, 10, 1, dof=None)
    eng = model.engine(inputs)
    eng.alloc_history(2100)
    for i in range(50): eng.train_step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(50, 2050): eng.train_step(i)
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    print(f"N={N}: host issue {1e6*t_issue/2000:.1f} us/step, wall {1e6*t_all/2000:.1f} us/step", flush=True)
