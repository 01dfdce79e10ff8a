"""The oracle against the committed golden vectors (tests/golden/*.npz, generator tests/golden/make_golden.py): loss, NLL,
KL, predictions, every gradient and an Adam trajectory.  CPU only.  The GPU twin of this test is in test_gpu_parity.py."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests import util
from tests.golden.make_golden import CASES, STEPS

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load_case(path):
    z = np.load(path)
    name = os.path.splitext(os.path.basename(path))[0]
    kw = CASES[name]
    if "data_refl_id" in z.files:                 # later files carry the whole problem
        data = {k[5:]: (z[k] if z[k].ndim else z[k].item()) for k in z.files if k.startswith("data_")}
    else:
        data = dict(refl_id=z["inputs_refl_id"], image_id=z["inputs_image_id"], file_id=z["inputs_file_id"],
                    metadata=z["inputs_metadata"], iobs=z["inputs_intensities"], sigiobs=z["inputs_uncertainties"],
                    centric=z["centric"], multiplicity=z["multiplicity"], n_images=int(z["n_images"]), n_refl=len(z["centric"]))
    image_layers = kw.get("image_layers", 0)
    cfg = O.ElboConfig(mc_samples=kw["S"], likelihood=kw.get("likelihood", "normal"), dof=kw.get("dof"),
                       scale_bijector=kw.get("bijector", "exp"), scale_shift=kw.get("shift", 0.0),
                       use_image_scales=kw.get("use_image_scales", True) and image_layers == 0, kl_weight=kw.get("kl_weight"),
                       prior="double_wilson" if kw.get("double_wilson") else "wilson", laue=kw.get("laue", False),
                       ev11=kw.get("ev11", False), optimize_dw_r=kw.get("optimize_dw_r", False), image_layers=image_layers,
                       clipnorm=kw.get("clipnorm"), clipvalue=kw.get("clipvalue"), global_clipnorm=kw.get("global_clipnorm"))
    n_t = len([k for k in z.files if k.startswith("param_")])
    ts = [torch.as_tensor(z[f"param_{i:02d}"].astype(np.float64)) for i in range(n_t)]
    n_layers = kw["L"] + 1
    mlp = ts[2:2 + 2 * n_layers]
    rest = ts[2 + 2 * n_layers:]                  # in ElboParams.tensors() order: image scales, image layers, Ev11, double-Wilson r
    img = rest.pop(0) if cfg.use_image_scales else None
    imgl = [rest.pop(0) for _ in range(2 * image_layers)]
    ev = rest.pop(0) if cfg.ev11 else None
    dwr = rest.pop(0) if cfg.optimize_dw_r else None
    assert not rest
    params = O.ElboParams(ts[0], ts[1], mlp[0::2], mlp[1::2], img, dwr, ev, imgl[0::2] or None, imgl[1::2] or None)
    return z, kw, data, cfg, params


def test_golden_files_exist():
    assert len(FILES) == len(CASES) and len(FILES) >= 8


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(path):
    z, kw, data, cfg, params = load_case(path)
    x = O.inputs_from_numpy(data)
    out, grads = O.elbo_value_and_grads(params, x, cfg, torch.as_tensor(z["u_f"], dtype=torch.float64),
                                        torch.as_tensor(z["eta"], dtype=torch.float64))
    for k in ("loss", "nll", "kl"):
        assert np.isclose(float(out[k]), float(z[k]), rtol=1e-10), k
    assert np.allclose(out["ipred"].numpy(), z["ipred"], rtol=1e-9, atol=1e-12)
    for i, g in enumerate(grads):
        assert np.allclose(g.numpy(), z[f"grad_{i:02d}"], rtol=1e-8, atol=1e-10), i
    p = params.clone()
    st = O.AdamState.zeros_like(p.tensors())
    hist = [O.train_step(p, x, cfg, st, torch.as_tensor(z["traj_u"][i], dtype=torch.float64),
                         torch.as_tensor(z["traj_eta"][i], dtype=torch.float64)) for i in range(STEPS)]
    assert np.allclose([h["loss"] for h in hist], z["traj_loss"], rtol=1e-9)
    assert np.allclose([h["Grad Norm"] for h in hist], z["traj_gnorm"], rtol=1e-8)
    for i, t in enumerate(p.tensors()):
        assert np.allclose(t.numpy(), z[f"final_{i:02d}"], rtol=1e-8, atol=1e-10)
