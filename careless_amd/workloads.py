"""Named workloads of BASELINE.json `configs` built from the synthetic generator: model + `inputs` tuple.

Shared by bench.py, `__graft_entry__.smoke()` and the tests so that all of them run the same configuration.
Model wiring follows `DataManager.build_model` (reference careless/io/manager.py:380-507): q initialised from the
Wilson prior's mean / stddev, `low = 1e-32 * ~centric`, identity-initialised MLP, image scales on (CLI default).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

from careless_amd.synthetic import make_synthetic

WORKLOADS: Dict[str, dict] = {
    # BASELINE.json configs[1]
    "mono_1M_normal_5x64_S1": dict(N=1_000_000, d0=5, posenc=False, L=5, w=64, S=1, dof=None, outliers=False),
    # BASELINE.json configs[2]: the configuration the headline metric is quoted on
    "mono_10M_studentt_posenc_5x64_S8": dict(N=10_000_000, d0=5, posenc=True, L=5, w=64, S=8, dof=16.0, outliers=True),
}


def flops_per_obs(d: int, w: int, L: int) -> int:
    """Algorithmic flops of the scaler per observation per step: 6 (d w + (L-1) w^2 + 2 w)  (SURVEY 8d)."""
    return 6 * (d * w + (L - 1) * w * w + 2 * w)


def bytes_per_obs(d: int, S: int, image_scales: bool = True) -> int:
    """Algorithmic HBM bytes per observation per step: 4 (d + 3) [+4 image_id] + 8 S  (SURVEY 8d)."""
    return 4 * (d + 3) + (4 if image_scales else 0) + 8 * S


def reference_inputs(data) -> Tuple[np.ndarray, ...]:
    """`inputs` in BaseModel.input_index order with the reference's shapes and dtypes (formatter.py:382-394)."""
    col = lambda a, t: np.asarray(a).astype(t)[:, None]
    return (col(data["refl_id"], np.int64), col(data["image_id"], np.int64), col(data["file_id"], np.int64),
            np.asarray(data["metadata"], dtype=np.float32), col(data["iobs"], np.float32), col(data["sigiobs"], np.float32))


def build_model(data, L: int, w: int, S: int, dof: Optional[float] = None, image_scales: bool = True,
                scale_bijector: str = "exp", epsilon: float = 1e-7, init_scale: float = 1.0, seed: int = 1234):
    from careless_amd.models.likelihoods.mono import NormalLikelihood, StudentTLikelihood
    from careless_amd.models.merging.surrogate_posteriors import TruncatedNormal
    from careless_amd.models.merging.variational import VariationalMergingModel
    from careless_amd.models.priors.wilson import WilsonPrior
    from careless_amd.models.scaling.image import HybridImageScaler, ImageScaler
    from careless_amd.models.scaling.nn import MLPScaler
    from careless_amd.optimizers import Adam

    prior = WilsonPrior(data["centric"], data["multiplicity"], 1.0)
    low = (1e-32 * ~np.asarray(data["centric"], dtype=bool)).astype(np.float32)          # manager.py:434
    q = TruncatedNormal.from_loc_and_scale(prior.mean(), prior.stddev() * init_scale, low, scale_shift=epsilon)
    lik = NormalLikelihood() if dof is None else StudentTLikelihood(dof)
    istd = float(np.asarray(data["iobs"]).std()) if scale_bijector == "softplus" else None  # manager.py:457
    mlp = MLPScaler(L, w, epsilon=epsilon, scale_bijector=scale_bijector, scale_multiplier=istd)
    scaler = HybridImageScaler(mlp, ImageScaler(int(data["n_images"]))) if image_scales else mlp
    model = VariationalMergingModel(q, prior, lik, scaler, mc_sample_size=S)
    model.seed = seed
    model.compile(Adam(1e-3, 0.9, 0.99))                                                  # args/optimizer.py
    return model


def make_workload(name: str, N: Optional[int] = None, seed: int = 1234):
    """Returns (model, inputs, data, spec) for a named workload; `N` overrides the observation count (bounded samples)."""
    spec = dict(WORKLOADS[name])
    if N is not None:
        spec["N"] = int(N)
    data = make_synthetic(spec["N"], d0=spec["d0"], posenc=spec["posenc"], outliers=spec["outliers"], seed=seed)
    model = build_model(data, spec["L"], spec["w"], spec["S"], dof=spec["dof"])
    spec["d"] = int(np.asarray(data["metadata"]).shape[1])
    spec["R"] = int(data["n_refl"])
    spec["M"] = int(data["n_images"])
    return model, reference_inputs(data), data, spec
