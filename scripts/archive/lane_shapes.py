"""Step time of the CLI-default scaler (20 layers) over metadata widths / hidden widths / MC samples: one line per shape.
SHAPES="w:d:S,..." (default: the shapes round 3 added to the lane kernel); N observations (default 4 M)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, ".")
from careless_amd.workloads import build_model, reference_inputs
from careless_amd.synthetic import make_synthetic
N = int(os.environ.get("N", "4000000"))
L = int(os.environ.get("LAYERS", "20"))
shapes = [tuple(int(v) for v in x.split(":")) for x in os.environ.get("SHAPES", "10:5:1,10:5:8,10:21:1,10:21:8,10:31:1,10:5:12,8:21:1,13:5:1,15:15:1").split(",")]
cache = {}
for w, d, S in shapes:
    if d not in cache:
        cache.clear()
        cache[d] = make_synthetic(N, d0=d, posenc=False, outliers=True)
    data = cache[d]
    model = build_model(data, L, w, S, dof=16.0)
    eng = model.engine(reference_inputs(data)); eng.alloc_history(30)
    for i in range(5): eng.train_step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): eng.train_step(5 + i)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
    F = 6 * (d * w + (L - 1) * w * w + 2 * w)
    print("mono %dM %dx%d d=%d S=%d studentt: %.3f ms/step %.3e refl/s mfma_frac(step) %.3f  %s  env LANE=%s NARROW=%s W4=%s" % (
        N // 1000000, L, w, d, S, 1e3 * t, N / t, F * N / t / 157.3e12, eng.kernel_name(),
        os.environ.get("CARELESS_HIP_LANE", "1"), os.environ.get("CARELESS_HIP_NARROW", "1"), os.environ.get("CARELESS_HIP_NARROW_W4", "0")), flush=True)
    del eng, model
