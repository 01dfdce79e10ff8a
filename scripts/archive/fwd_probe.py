#!/usr/bin/env python3
"""Diagnostic: time of the forward-only launch (cl_mlp_forward) of the fused scaler kernel on the headline geometry.
Usage: CARELESS_HIP_LIB=... python scripts/fwd_probe.py [nobs] [d] [L] [w]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from careless_amd import _lib
from careless_amd._lib import MlpArgs, ptr, check
from careless_amd.models.scaling.nn import MLPScaler
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 21
L = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w = int(sys.argv[4]) if len(sys.argv) > 4 else 64
lib = _lib.get_lib()
dev = torch.device("cuda")
mlp = MLPScaler(L, w, scale_bijector="exp"); mlp.build(d); mlp.flat = (mlp.flat + 0.05 * torch.randn_like(mlp.flat)).to(dev)
n_pad = (N + 127) // 128 * 128
meta = torch.randn(int(lib.cl_mlp_meta_rows(d)), n_pad, device=dev)
loc = torch.empty(N, device=dev); sig = torch.empty(N, device=dev)
a = MlpArgs()
a.meta_t, a.n_obs, a.n_pad = ptr(meta), N, n_pad
a.mlp = ptr(mlp.flat); a.d, a.w, a.L, a.leak = d, w, L, 0.01
a.bij_kind, a.eps, a.S, a.R = 0, 1e-7, 1, 1
a.loc_out, a.sig_out = ptr(loc), ptr(sig)
grid = int(lib.cl_mlp_default_grid())
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    check(lib.cl_mlp_forward(C.byref(a), grid, st), "fwd")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    check(lib.cl_mlp_forward(C.byref(a), grid, st), "fwd")
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
dp = 8 if d <= 8 else (32 if d <= 32 else 64)
wp = 16 if w <= 16 else (32 if w <= 32 else 64)
mfma = (dp // 4) * (wp // 16) + (L - 1) * (wp // 16) ** 2 * 4          # per wave-tile
pipe_ms = (n_pad / 128) / grid * 2 * mfma * 32 / 2.4e9 * 1e3
print(f"{os.environ.get('CARELESS_HIP_LIB', 'default'):40s} fwd-only N={N} d={d} {L}x{w}: {ms:.3f} ms   MFMA-pipe bound {pipe_ms:.3f} ms  -> {100 * pipe_ms / ms:.1f}%  checksum {float(loc.sum()):.6g}")
