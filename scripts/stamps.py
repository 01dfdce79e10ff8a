#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fused kernel (needs `python -m careless_amd.build --stamps`).
Usage: python scripts/stamps.py [workload] [nobs]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from careless_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libcareless_hip_stamps.so")
import torch
from careless_amd.workloads import make_workload
wl = sys.argv[1] if len(sys.argv) > 1 else "mono_10M_studentt_posenc_5x64_S8"
nobs = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
model, inputs, data, spec = make_workload(wl, N=nobs)
eng = model.engine(inputs)
NPH = 16
dbg = torch.zeros(eng.grid * 8 * NPH, dtype=torch.int64, device=eng.device)
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
d = dbg.view(eng.grid, 8, NPH).cpu().numpy().astype(np.float64)
names = ["load(h0)+vmwait", "fwd", "prefetch issue", "seam barrier", "top: dH init + Dense2 wgrad", "dZ", "barrier A", "stage writes + bias",
         "barrier B", "wgrad", "dgrad(copy)", "epi: dense2 combine+bijector", "epi: sample loop", "epi: reduce+img atomic",
         "epi: tail before prefetch", "-"]
tot = d.sum(-1).mean()
tiles = (eng.n_pad // 128) / eng.grid
print(f"workload {wl} nobs {nobs}: mean cycles per wave {tot:.0f}, per tile {tot / tiles:.0f} (100 MHz ticks x? see note)")
for k, n in enumerate(names):
    v = d[:, :, k].mean()
    print(f"  {n:32s} {v / tiles:10.1f} /tile  {100 * v / tot:5.1f}%   (waves0-3 {d[:, :4, k].mean() / tiles:9.1f}, waves4-7 {d[:, 4:, k].mean() / tiles:9.1f})")
