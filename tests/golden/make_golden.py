#!/usr/bin/env python3
"""Generator of the committed golden vectors (tests/golden/*.npz).

The reference (TF/TFP) cannot be imported in the build container, so these vectors are produced by the fp64 CPU oracle
(oracle/elbo_oracle.py) after it has been pinned against the reference's closed-form known-answer tests and scipy
(tests/test_oracle_kat.py).  They freeze the oracle: any later edit of the oracle that changes a loss, a gradient or an
Adam trajectory fails tests/test_golden.py.  The `inputs_*` arrays are stored in `BaseModel.input_index` order so the
same file can be replayed through the real reference if a machine with TensorFlow ever becomes available.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import elbo_oracle as O  # noqa: E402
from tests import util  # noqa: E402

CASES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cases.json")))     # shared with scripts/replay_golden_in_reference.py
FROZEN = ("mono_2x32_normal_S3", "mono_5x64_studentt_posenc_S8", "mono_3x20_softplus_shift_noimg_S2", "mono_2x16_klweight_S4")
STEPS = 6


def extreme_uniforms(u_f):
    """A few injected uniforms at the upper end of (0, 1): the largest float32 below one, 1 - 2^-24 > 1 - eps_f32, so that the
    [tiny, 1 - eps] clip of TFP's truncated-normal sample gradient (oracle._TNStdSample, [3P-recall]) is ACTIVE in the case -- with
    loc = exp(a) > 0 >= low the truncation point never lies in the upper tail, so the end of u is the only way to reach the clip.
    (The lower end is left alone: a uniform of 1e-30 or 2^-24 puts the sample within 1e-6 of the truncation point, where
    z = loc + scale e cancels in float32 and log p(z) ~ log z is singular -- an fp32 engine and an fp64 oracle differ there by
    rounding alone, as fp32 TensorFlow would; `tiny` is out of reach of any float32 uniform anyway.)"""
    u = np.array(u_f, dtype=np.float32)
    u[0, ::5] = np.nextafter(np.float32(1.0), np.float32(0.0))
    u[-1, 1::5] = np.nextafter(np.float32(1.0), np.float32(0.0))
    return u


def make(name, kw):
    kw = dict(kw)
    extreme = kw.pop("extreme_u", False)
    data, cfg, params, x, u_f, eta = util.make_problem(**kw)
    if extreme:
        u_f = extreme_uniforms(u_f)
    out, grads = O.elbo_value_and_grads(params, x, cfg, torch.as_tensor(u_f, dtype=torch.float64),
                                        torch.as_tensor(eta, dtype=torch.float64))
    rng = np.random.default_rng(99)
    S, R, N = kw["S"], kw["R"], kw["N"]
    noises_u = rng.random((STEPS, S, R)).astype(np.float32)
    if extreme:
        noises_u = np.stack([extreme_uniforms(u) for u in noises_u])
    noises_e = rng.normal(size=(STEPS, S, N)).astype(np.float32)
    p = params.clone()
    st = O.AdamState.zeros_like(p.tensors())
    hist = [O.train_step(p, x, cfg, st, torch.as_tensor(noises_u[i], dtype=torch.float64),
                         torch.as_tensor(noises_e[i], dtype=torch.float64)) for i in range(STEPS)]
    arrs = dict(
        inputs_refl_id=np.asarray(data["refl_id"]), inputs_image_id=np.asarray(data["image_id"]),
        inputs_file_id=np.asarray(data["file_id"]), inputs_metadata=np.asarray(data["metadata"]),
        inputs_intensities=np.asarray(data["iobs"]), inputs_uncertainties=np.asarray(data["sigiobs"]),
        centric=np.asarray(data["centric"]), multiplicity=np.asarray(data["multiplicity"]),
        n_images=np.int64(data["n_images"]), u_f=u_f, eta=eta,
        loss=np.float64(out["loss"]), nll=np.float64(out["nll"]), kl=np.float64(out["kl"]),
        ipred=out["ipred"].numpy(), z_f=out["z_f"].numpy(),
        traj_u=noises_u, traj_eta=noises_e,
        traj_loss=np.array([h["loss"] for h in hist]), traj_gnorm=np.array([h["Grad Norm"] for h in hist]),
        traj_kl=np.array([h["F KLDiv"] for h in hist]), traj_nll=np.array([h["NLL"] for h in hist]),
    )
    for k, v in data.items():                     # the whole problem (Laue / double-Wilson arrays included)
        arrs[f"data_{k}"] = np.asarray(v)
    for i, t in enumerate(params.tensors()):
        arrs[f"param_{i:02d}"] = t.numpy().astype(np.float32)
    for i, g in enumerate(grads):
        arrs[f"grad_{i:02d}"] = g.numpy()
    for i, t in enumerate(p.tensors()):
        arrs[f"final_{i:02d}"] = t.numpy()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **arrs)
    print(name, float(out["loss"]))


if __name__ == "__main__":
    for name, kw in CASES.items():
        if name in FROZEN and "--all" not in sys.argv:
            continue
        make(name, kw)
