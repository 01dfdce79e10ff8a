"""Oracle-independent check (round 5): does the engine MERGE?  Train on synthetic data whose true amplitudes are known
(careless_amd/synthetic.py: f_true) with in-kernel noise and report the Pearson correlation of the merged F with F_true, the
correlation of the two half-dataset merges of `--merge-half-datasets`, and the loss at step 200 / at the end.  Calibrates the
thresholds of tests/test_recovery.py.   python scripts/recovery_check.py [kind=mono|laue|dw] [N] [steps] [lr ...]
Reference flow: careless/careless.py:61-128, io/manager.py:188-197."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from careless_amd.manager import DataManager, default_args, merge_half_datasets
from careless_amd.synthetic import make_synthetic, make_synthetic_double_wilson, make_synthetic_laue
from careless_amd.workloads import reference_inputs


def cc(a, b):
    return float(np.corrcoef(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))[0, 1])


def problem(kind, N):
    if kind == "laue":
        d = make_synthetic_laue(N)
        col = lambda a, t: np.asarray(a).astype(t)[:, None]
        inputs = reference_inputs(d) + (col(d["wavelength"], np.float32), col(d["harmonic_id"], np.int64))
        return d, inputs, dict(type="poly"), None
    if kind == "dw":
        d = make_synthetic_double_wilson(N)
        dw = dict(reflids=d["parent_ids"], root=d["root"], asu_ids=d["asu_ids"])
        return d, reference_inputs(d), dict(parents="None,0", dwr="0.,0.9"), dw
    d = make_synthetic(N)
    return d, reference_inputs(d), {}, None


def run(kind, N, steps, lr, halves=True, L=5, w=64):
    d, inputs, extra, dw = problem(kind, N)
    args = default_args(mlp_layers=L, mlp_width=w, iterations=steps, learning_rate=lr, **extra)
    np.random.seed(args.seed)
    dm = DataManager(inputs, d["centric"], d["multiplicity"], parser=args, double_wilson=dw)
    model = dm.build_model()
    t0 = time.time()
    hist = model.train_model(dm.inputs, steps, progress=False)
    t1 = time.time()
    res = dm.get_results(model.surrogate_posterior)
    obs = res["observed"]
    out = dict(kind=kind, N=N, steps=steps, lr=lr, seconds=round(t1 - t0, 2), cc_true=cc(res["F"][obs], d["f_true"][obs]),
               cc_true_I=cc(res["I"][obs], d["f_true"][obs] ** 2), loss_200=hist["loss"][min(200, steps - 1)], loss_end=hist["loss"][-1],
               finite=bool(np.all(np.isfinite(hist["loss"]))))
    if halves:
        hv = merge_half_datasets(dm, args, model.scaling_model, steps)
        (_, _, a), (_, _, b) = hv[0], hv[1]
        both = a["observed"] & b["observed"]
        out.update(cc_half=cc(a["F"][both], b["F"][both]), n_both=int(both.sum()),
                   cc_half_true=(cc(a["F"][a["observed"]], d["f_true"][a["observed"]]), cc(b["F"][b["observed"]], d["f_true"][b["observed"]])))
    return out


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "mono"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    lrs = [float(x) for x in sys.argv[4:]] or [1e-3, 1e-2]
    for lr in lrs:
        print(run(kind, N, steps, lr), flush=True)
