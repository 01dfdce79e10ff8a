"""Shared helpers of the parity tests: build the same problem for the oracle (fp64, CPU) and for the HIP engine."""
from __future__ import annotations

import numpy as np
import torch

from oracle import elbo_oracle as O


def make_problem(N=300, R=40, d0=5, posenc=False, n_images=4, L=2, w=32, S=3, likelihood="normal", dof=None,
                 bijector="exp", shift=0.0, use_image_scales=True, kl_weight=None, perturb=0.05, seed=7,
                 outliers=False, double_wilson=False, laue=False, ev11=False, optimize_dw_r=False, image_layers=0, extra_meta=0, **opt):
    if image_layers > 0:
        use_image_scales = False        # NeuralImageScaler replaces the HybridImageScaler (manager.py:467-489)
        opt["image_layers"] = image_layers
    if laue:
        data = O.make_synthetic_laue(N, R=R, n_images=n_images, seed=seed)
    elif double_wilson:
        data = O.make_synthetic_double_wilson(N, R_half=R // 2, d0=d0, posenc=posenc, n_images=n_images, seed=seed,
                                              outliers=outliers)
    else:
        data = O.make_synthetic(N, R=R, d0=d0, posenc=posenc, n_images=n_images, seed=seed, outliers=outliers)
    if extra_meta:                      # more metadata columns than the generator makes (Laue data with positional encodings)
        more = np.random.default_rng(seed + 5).normal(size=(N, extra_meta)).astype(np.float32)
        data["metadata"] = np.concatenate([np.asarray(data["metadata"], dtype=np.float32), more], axis=1)
    cfg = O.ElboConfig(mc_samples=S, likelihood=likelihood, dof=dof, scale_bijector=bijector, scale_shift=shift,
                       use_image_scales=use_image_scales, kl_weight=kl_weight,
                       prior="double_wilson" if double_wilson else "wilson", laue=laue, ev11=ev11, optimize_dw_r=optimize_dw_r, **opt)
    rng = np.random.default_rng(seed + 1)
    params = O.init_params(data, cfg, L, w, perturb=perturb, rng=rng)
    x = O.inputs_from_numpy(data)
    u_f = rng.random((S, R)).astype(np.float32)
    eta = rng.normal(size=(S, N)).astype(np.float32)
    return data, cfg, params, x, u_f, eta


def reference_inputs(data):
    """The `inputs` tuple in BaseModel.input_index order with the reference's shapes/dtypes (formatter.py:382-394)."""
    col = lambda a, t: np.asarray(a).astype(t)[:, None]
    tup = (col(data["refl_id"], np.int64), col(data["image_id"], np.int64), col(data["file_id"], np.int64),
           np.asarray(data["metadata"], dtype=np.float32), col(data["iobs"], np.float32), col(data["sigiobs"], np.float32))
    if "harmonic_id" in data:
        tup = tup + (col(data["wavelength"], np.float32), col(data["harmonic_id"], np.int64))
    return tup


def build_model(data, cfg: O.ElboConfig, params: O.ElboParams, L, w):
    """careless_amd model holding exactly the oracle's parameters."""
    from careless_amd.models.likelihoods.mono import NormalLikelihood, StudentTLikelihood
    from careless_amd.models.merging.surrogate_posteriors import TruncatedNormal
    from careless_amd.models.merging.variational import VariationalMergingModel
    from careless_amd.models.priors.wilson import DoubleWilsonPrior, WilsonPrior
    from careless_amd.models.scaling.image import HybridImageScaler, ImageScaler, NeuralImageScaler
    from careless_amd.models.scaling.nn import MLPScaler
    from careless_amd.optimizers import Adam

    if cfg.prior == "double_wilson":
        prior = DoubleWilsonPrior(data["centric"], data["multiplicity"], data["parent_ids"], data["root"], data["asu_ids"],
                                  data["dw_r"], parents=[None, 0], optimize_r=cfg.optimize_dw_r)
        if cfg.optimize_dw_r:
            prior.r_raw = torch.as_tensor(params.dw_r_raw.numpy().astype(np.float32))
    else:
        prior = WilsonPrior(data["centric"], data["multiplicity"], 1.0)
    low = (1e-32 * ~np.asarray(data["centric"], dtype=bool)).astype(np.float32)
    q = TruncatedNormal(params.q_loc_raw.numpy().astype(np.float32), params.q_scale_raw.numpy().astype(np.float32),
                        low, high=cfg.high, scale_shift=cfg.epsilon)
    from careless_amd.models.likelihoods import laue as laue_lik, mono as mono_lik
    mod = laue_lik if cfg.laue else mono_lik
    if cfg.ev11:
        lik = mod.NormalEv11Likelihood() if cfg.likelihood == "normal" else mod.StudentTEv11Likelihood(cfg.dof)
        lik.raw = torch.as_tensor(params.ev11_raw.numpy().astype(np.float32))
    else:
        lik = mod.NormalLikelihood() if cfg.likelihood == "normal" else mod.StudentTLikelihood(cfg.dof)
    d = np.asarray(data["metadata"]).shape[1]
    nis = None
    if cfg.image_layers > 0:
        nis = NeuralImageScaler(cfg.image_layers, int(data["n_images"]), L, w, leakiness=cfg.leakiness, epsilon=cfg.epsilon,
                                scale_bijector=cfg.scale_bijector, scale_multiplier=(cfg.scale_shift if cfg.scale_shift else None))
        nis.build(d)
        mlp = nis.metadata_scaler
        for dst, wt, b in zip(range(cfg.image_layers), params.imgl_w, params.imgl_b):
            nis.image_weights[2 * dst].copy_(torch.as_tensor(wt.numpy().astype(np.float32)))
            nis.image_weights[2 * dst + 1].copy_(torch.as_tensor(b.numpy().astype(np.float32)))
    else:
        mlp = MLPScaler(L, w, leakiness=cfg.leakiness, epsilon=cfg.epsilon, scale_bijector=cfg.scale_bijector,
                        scale_multiplier=(cfg.scale_shift if cfg.scale_shift else None))
    mlp.build(d)
    ws = []
    for wt, b in zip(params.mlp_w, params.mlp_b):
        ws += [wt.numpy().astype(np.float32), b.numpy().astype(np.float32)]
    mlp.set_weights(ws)
    if cfg.use_image_scales:
        img = ImageScaler(int(data["n_images"]))
        img._scales.copy_(torch.as_tensor(params.img_raw.numpy().astype(np.float32)))
        scaler = HybridImageScaler(mlp, img)
    elif nis is not None:
        scaler = nis
    else:
        scaler = mlp
    model = VariationalMergingModel(q, prior, lik, scaler, mc_sample_size=cfg.mc_samples, kl_weight=cfg.kl_weight)
    model.compile(Adam(cfg.learning_rate, cfg.beta_1, cfg.beta_2, cfg.adam_epsilon, clipnorm=cfg.clipnorm,
                       clipvalue=cfg.clipvalue, global_clipnorm=cfg.global_clipnorm))
    return model


def rel_err(a, b):
    """max |a-b| / max|b| -- tensor-level relative error."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b)) / den)
