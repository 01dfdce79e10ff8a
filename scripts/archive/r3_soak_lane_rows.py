"""Trajectory consistency of the round-3 lane kernel (metadata rows in LDS, 21 columns) against the fused 16-wide instance the same shape
ran on before (CARELESS_HIP_LANE=0, read once per process: two child processes): 600 Adam steps on a 200 k-observation problem with
in-kernel noise (same keys), Student-T, 3 MC samples.  Prints both loss curves at a few steps and the relative difference of the final
parameters."""
import os, subprocess, sys, json
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ".")
    import torch
    from careless_amd.workloads import build_model, reference_inputs
    from careless_amd.synthetic import make_synthetic
    data = make_synthetic(200_000, d0=5, posenc=True, outliers=True)
    model = build_model(data, 20, 10, 3, dof=16.0)
    h = model.train_model(reference_inputs(data), 600, progress=False)
    eng = model._engine
    np.save(sys.argv[2], eng.params.cpu().numpy())
    print(json.dumps({"kernel": eng.kernel_name(), "loss": [h["loss"][i] for i in (0, 1, 10, 100, 300, 599)]}))
    sys.exit(0)

outs = []
for lane in ("1", "0"):
    env = dict(os.environ, CARELESS_HIP_LANE=lane, CARELESS_HIP_NARROW="0" if lane == "0" else "1")
    f = f"/tmp/soak_params_{lane}.npy"
    r = subprocess.run([sys.executable, __file__, "child", f], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(r.stderr[-2000:]); sys.exit(1)
    d = json.loads(line[-1]); d["params"] = np.load(f); outs.append(d)
    print(d["kernel"], ["%.6e" % v for v in d["loss"]])
a, b = outs[0]["params"], outs[1]["params"]
print("final parameters: max |a - b| / max |b| = %.3e ; losses differ by %.3e relative at step 599" % (
    np.abs(a - b).max() / np.abs(b).max(), abs(outs[0]["loss"][-1] - outs[1]["loss"][-1]) / abs(outs[1]["loss"][-1])))
