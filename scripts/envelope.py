"""The default scaler's neighbourhood as a table (round 5): fraction of the fp32 matrix rate on STEP time and the kernel
the library routes to, over depth L, hidden width w, metadata columns d and MC samples S.  4 M observations, Student-T, image scales on,
in-kernel noise; 10 timed steps per cell.  Reference flags: careless/args/scaling.py:21-31, args/positional_encoding.py:24-37.
    python scripts/envelope.py > profiles/r5_envelope.txt       env: N, LS, WS, DS, SS (comma lists), IMGL (per-image layers on top: `--image-layers`)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from careless_amd.synthetic import make_synthetic
from careless_amd.workloads import build_model, reference_inputs

lst = lambda k, d: [int(v) for v in os.environ.get(k, d).split(",")]
N = int(os.environ.get("N", "4000000"))
IMGL = int(os.environ.get("IMGL", "0"))
LS, WS, DS, SS = lst("LS", "5,10,12,20,24"), lst("WS", "4,8,10,12,15"), lst("DS", "5,15,21,31,37,53"), lst("SS", "1,8")
print(f"# N = {N} observations, Student-T(16), image scales, in-kernel noise" + (f", {IMGL} per-image layers" if IMGL else "") + f"; cell = MFMA fraction on step time (ms per step) kernel", flush=True)
for d in DS:
    data = make_synthetic(N, d0=d, posenc=False, outliers=True)
    inputs = reference_inputs(data)
    for L in LS:
        for S in SS:
            cells = []
            for w in WS:
                try:
                    model = build_model(data, L, w, S, dof=16.0, image_layers=IMGL)
                    eng = model.engine(inputs)
                    eng.alloc_history(16)
                    for i in range(3):
                        eng.train_step(i)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(10):
                        eng.train_step(3 + i)
                    torch.cuda.synchronize()
                    t = (time.perf_counter() - t0) / 10
                    F = 6 * (d * w + (L - 1 + IMGL) * w * w + 2 * w)
                    name = eng.kernel_name().split("<")[0].replace("elbo_", "").replace("_kernel", "")
                    if eng.blocks is not None:
                        name += f"x{len(eng.blocks)}"
                    cells.append("w=%-2d %.3f (%.3f ms) %-7s" % (w, F * N / t / 157.3e12, 1e3 * t, name))
                    del eng, model
                except Exception as e:       # noqa: BLE001
                    cells.append("w=%-2d FAILED %r" % (w, e))
            print("d=%-2d L=%-2d S=%d | " % (d, L, S) + " | ".join(cells), flush=True)
    del data, inputs
