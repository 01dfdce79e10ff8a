#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the lane-per-observation kernel (library built with -DCL_STAMPS: scripts/build_lane_exp.sh stamps -DCL_STAMPS).
Usage: CARELESS_HIP_LIB=careless_amd/lib/exp_stamps.so python scripts/stamps_lane.py [workload] [nobs]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from careless_amd.workloads import make_workload
wl = sys.argv[1] if len(sys.argv) > 1 else "mono_10M_cli_default_20x10_S1"
nobs = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
NWV = 4
model, inputs, data, spec = make_workload(wl, N=nobs)
eng = model.engine(inputs)
dbg = torch.zeros(eng.grid * NWV * 8, dtype=torch.int64, device=eng.device)
orig = eng._mlp_args
def patched(step, eta, ipred_out=None, obs=None):
    a = orig(step, eta, ipred_out, obs)
    a.loc_out = dbg.data_ptr()
    return a
eng._mlp_args = patched
eng.alloc_history(4)
for i in range(3):
    eng.train_step(i)
torch.cuda.synchronize()
d = dbg.view(eng.grid, NWV, 8).cpu().numpy().astype(np.float64)
names = ["tile prologue (inputs, gathers)", "forward layers", "head + epilogue", "prefetch issue", "backward head", "backward layers", "LAUNCH prologue (images, fill)", "LAUNCH flush"]
tot = d.sum(-1).mean()
tiles = -(-nobs // 64) / (eng.grid * NWV)
print(f"{eng.kernel_name()} workload {wl} nobs {nobs}: wave tiles per wave {tiles:.1f}, mean ticks per wave {tot:.0f}, per wave tile {tot / tiles / 3:.0f} (3 launches accumulated)")
for k, n in enumerate(names):
    v = d[:, :, k].mean()
    print(f"  {n:34s} {v / tiles / 3:10.1f} /tile  {100 * v / tot:5.1f}%   per launch {v / 3:10.0f} ticks")
