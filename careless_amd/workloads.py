"""Named workloads of BASELINE.json `configs` built from the synthetic generator: model + `inputs` tuple.

Shared by bench.py, `__graft_entry__.smoke()` and the tests so that all of them run the same configuration.
Model wiring follows `DataManager.build_model` (reference careless/io/manager.py:380-507): q initialised from the
Wilson prior's mean / stddev, `low = 1e-32 * ~centric`, identity-initialised MLP, image scales on (CLI default).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

from careless_amd.synthetic import make_synthetic, make_synthetic_double_wilson, make_synthetic_laue

WORKLOADS: Dict[str, dict] = {
    # BASELINE.json configs[1]
    "mono_1M_normal_5x64_S1": dict(N=1_000_000, d0=5, posenc=False, L=5, w=64, S=1, dof=None, outliers=False),
    # BASELINE.json configs[2]: the configuration the headline metric is quoted on
    "mono_10M_studentt_posenc_5x64_S8": dict(N=10_000_000, d0=5, posenc=True, L=5, w=64, S=8, dof=16.0, outliers=True),
    # BASELINE.json configs[3]: Laue harmonic deconvolution (quoted on 4 GPUs; runs on one)
    "laue_5M_normal_5x64_S1": dict(N=5_000_000, d0=5, posenc=False, L=5, w=64, S=1, dof=None, outliers=False, kind="laue"),
    # BASELINE.json configs[4]: two-ASU double-Wilson prior (quoted on 8 GPUs; runs on one)
    "dw_50M_normal_5x64_S1": dict(N=50_000_000, d0=5, posenc=False, L=5, w=64, S=1, dof=None, outliers=False, kind="double_wilson"),
    # the careless CLI defaults: --mlp-layers 20 --mlp-width 10 (args/scaling.py), Normal likelihood, mc-samples 1
    "mono_10M_cli_default_20x10_S1": dict(N=10_000_000, d0=5, posenc=False, L=20, w=10, S=1, dof=None, outliers=False),
    # configs[2]'s data (Student-T, positional encodings of X, Y: 5 + 16 metadata columns, mc-samples 8) on the CLI-default scaler:
    # what `careless mono --positional-encoding-keys X,Y --studentt-likelihood-dof 16 --mc-samples 8` runs
    "mono_10M_studentt_posenc_20x10_S8": dict(N=10_000_000, d0=5, posenc=True, L=20, w=10, S=8, dof=16.0, outliers=True),
    # FOUR positionally encoded keys (5 + 32 = 37 metadata columns) on the CLI-default scaler: past the lane kernel's 31 columns, the
    # first layer is peeled (csrc/elbo_peel.hip, round 5)
    "mono_10M_studentt_posenc4_20x10_S8": dict(N=10_000_000, d0=5, posenc=True, posenc_keys=4, L=20, w=10, S=8, dof=16.0, outliers=True),
    # a scaler wider than the fused kernels hold (hidden width > 64): layer-by-layer GEMM kernels (csrc/wide_gemm.hip)
    "mono_2M_studentt_3x128_S4": dict(N=2_000_000, d0=5, posenc=False, L=3, w=128, S=4, dof=16.0, outliers=True),
    # half the default depth (register-pressure experiments on the narrow kernel)
    "mono_10M_10x10_S1": dict(N=10_000_000, d0=5, posenc=False, L=10, w=10, S=1, dof=None, outliers=False),
    # (round 6) more layers than one launch holds at the default width: 4 layers on the 16-wide kernel's chain modes + 20 on the lane kernel
    "mono_10M_24x10_S1": dict(N=10_000_000, d0=5, posenc=False, L=24, w=10, S=1, dof=None, outliers=False),
    # --image-layers 1 on the headline configuration (one Dense layer traded for a per-image layer)
    "mono_10M_studentt_posenc_4x64_img1_S8": dict(N=10_000_000, d0=5, posenc=True, L=4, w=64, S=8, dof=16.0, outliers=True,
                                                  image_layers=1),
    # the CLI-default scaler on the other data kinds a user runs it on (round 5: measured so that the table has no unmeasured default)
    "laue_5M_normal_20x10_S1": dict(N=5_000_000, d0=5, posenc=False, L=20, w=10, S=1, dof=None, outliers=False, kind="laue"),
    "dw_10M_normal_20x10_S1": dict(N=10_000_000, d0=5, posenc=False, L=20, w=10, S=1, dof=None, outliers=False, kind="double_wilson"),
    "mono_10M_20x10_img2_S1": dict(N=10_000_000, d0=5, posenc=False, L=20, w=10, S=1, dof=None, outliers=False, image_layers=2),
    "mono_10M_studentt_posenc_20x10_img2_S8": dict(N=10_000_000, d0=5, posenc=True, L=20, w=10, S=8, dof=16.0, outliers=True, image_layers=2),
    # (round 6) `--image-layers 3`: the lane kernel's three-layer instance (a unit compiled without -amdgpu-mfma-vgpr-form)
    "mono_10M_20x10_img3_S1": dict(N=10_000_000, d0=5, posenc=False, L=20, w=10, S=1, dof=None, outliers=False, image_layers=3),
    # (round 6) `--mlp-layers 10 --image-layers 2`: the per-image-layer instance of the lane kernel's depth-10 unit
    "mono_10M_10x10_img2_S1": dict(N=10_000_000, d0=5, posenc=False, L=10, w=10, S=1, dof=None, outliers=False, image_layers=2),
    "laue_5M_normal_20x10_img2_S1": dict(N=5_000_000, d0=5, posenc=False, L=20, w=10, S=1, dof=None, outliers=False, kind="laue", image_layers=2),
}


def flops_per_obs(d: int, w: int, L: int, image_layers: int = 0) -> int:
    """Algorithmic flops of the scaler per observation per step: 6 (d w + (L-1) w^2 + 2 w)  (SURVEY 8d); a per-image layer
    costs the same w x w product as a Dense layer."""
    return 6 * (d * w + (L - 1 + image_layers) * w * w + 2 * w)


def bytes_per_obs(d: int, S: int, image_scales: bool = True) -> int:
    """Algorithmic HBM bytes per observation per step: 4 (d + 3) [+4 image_id] + 8 S  (SURVEY 8d)."""
    return 4 * (d + 3) + (4 if image_scales else 0) + 8 * S


def reference_inputs(data) -> Tuple[np.ndarray, ...]:
    """`inputs` in BaseModel.input_index order with the reference's shapes and dtypes (formatter.py:382-394)."""
    col = lambda a, t: np.asarray(a).astype(t, copy=False)[:, None]       # (views: the arrays may be memory maps shared by the ranks)
    return (col(data["refl_id"], np.int64), col(data["image_id"], np.int64), col(data["file_id"], np.int64),
            np.asarray(data["metadata"]).astype(np.float32, copy=False), col(data["iobs"], np.float32), col(data["sigiobs"], np.float32))


def build_model(data, L: int, w: int, S: int, dof: Optional[float] = None, image_scales: bool = True,
                scale_bijector: str = "exp", epsilon: float = 1e-7, init_scale: float = 1.0, seed: int = 1234,
                kind: str = "mono", image_layers: int = 0):
    from careless_amd.models.merging.surrogate_posteriors import TruncatedNormal
    from careless_amd.models.merging.variational import VariationalMergingModel
    from careless_amd.models.priors.wilson import DoubleWilsonPrior, WilsonPrior
    from careless_amd.models.scaling.image import HybridImageScaler, ImageScaler, NeuralImageScaler
    from careless_amd.models.scaling.nn import MLPScaler
    from careless_amd.optimizers import Adam
    if kind == "laue":
        from careless_amd.models.likelihoods.laue import NormalLikelihood, StudentTLikelihood
    else:
        from careless_amd.models.likelihoods.mono import NormalLikelihood, StudentTLikelihood

    if kind == "double_wilson":
        prior = DoubleWilsonPrior(data["centric"], data["multiplicity"], data["parent_ids"], data["root"], data["asu_ids"],
                                  data["dw_r"], parents=[None, 0])
    else:
        prior = WilsonPrior(data["centric"], data["multiplicity"], 1.0)
    low = (1e-32 * ~np.asarray(data["centric"], dtype=bool)).astype(np.float32)          # manager.py:434
    q = TruncatedNormal.from_loc_and_scale(prior.mean(), prior.stddev() * init_scale, low, scale_shift=epsilon)
    lik = NormalLikelihood() if dof is None else StudentTLikelihood(dof)
    istd = float(np.asarray(data["iobs"]).std()) if scale_bijector == "softplus" else None  # manager.py:457
    if image_layers > 0:                                                                   # manager.py:467-478
        scaler = NeuralImageScaler(image_layers, int(data["n_images"]), L, w, epsilon=epsilon, scale_bijector=scale_bijector,
                                   scale_multiplier=istd)
    else:
        mlp = MLPScaler(L, w, epsilon=epsilon, scale_bijector=scale_bijector, scale_multiplier=istd)
        scaler = HybridImageScaler(mlp, ImageScaler(int(data["n_images"]))) if image_scales else mlp
    model = VariationalMergingModel(q, prior, lik, scaler, mc_sample_size=S)
    model.seed = seed
    model.compile(Adam(1e-3, 0.9, 0.99))                                                  # args/optimizer.py
    return model


def _generate(spec: dict, seed: int) -> dict:
    kind = spec["kind"]
    if kind == "laue":
        return make_synthetic_laue(spec["N"], seed=seed)
    if kind == "double_wilson":
        return make_synthetic_double_wilson(spec["N"], d0=spec["d0"], posenc=spec["posenc"], outliers=spec["outliers"], seed=seed)
    return make_synthetic(spec["N"], d0=spec["d0"], posenc=spec["posenc"], outliers=spec["outliers"], seed=seed, posenc_keys=spec.get("posenc_keys", 2))


def _share_save(data: dict, path: str) -> None:
    """Rank 0: every array of the problem as a `.npy` file under `path` (a directory in /dev/shm), scalars in a manifest."""
    import json
    import os
    os.makedirs(path, exist_ok=True)
    scalars = {}
    for k, v in data.items():
        if isinstance(v, np.ndarray):
            np.save(os.path.join(path, k + ".npy"), v)
        else:
            scalars[k] = v if not isinstance(v, np.generic) else v.item()
    with open(os.path.join(path, "manifest.json.tmp"), "w") as f:
        json.dump(scalars, f)
    os.replace(os.path.join(path, "manifest.json.tmp"), os.path.join(path, "manifest.json"))


def _share_load(path: str) -> dict:
    """Any rank: the problem as read-only memory maps of rank 0's files (pages are shared between the ranks of the node)."""
    import json
    import os
    with open(os.path.join(path, "manifest.json")) as f:
        data = json.load(f)
    for fn in os.listdir(path):
        if fn.endswith(".npy"):
            data[fn[:-4]] = np.load(os.path.join(path, fn), mmap_mode="r")
    return data


def make_workload(name: str, N: Optional[int] = None, seed: int = 1234, rank: int = 0, world: int = 1, share_dir: Optional[str] = None,
                  barrier=None):
    """Returns (model, inputs, data, spec) for a named workload; `N` overrides the observation count (bounded samples).
    With `world > 1` and a `share_dir` (a directory every rank of the node sees, e.g. under /dev/shm) only rank 0 runs the
    generator -- its peak is ~250 B per observation -- and writes the arrays there; after `barrier()` every rank (rank 0 too: it
    drops its in-memory copy) maps the files read-only, so the node holds ONE copy of the inputs instead of `world` generator
    peaks (configs[4]: 50 M observations x 8 ranks).  The engine copies only its own shard's rows (`engine.ObsData`)."""
    spec = dict(WORKLOADS[name])
    if N is not None:
        spec["N"] = int(N)
    kind = spec.setdefault("kind", "mono")
    spec.setdefault("image_layers", 0)
    if world > 1 and share_dir is not None:
        if rank == 0:
            _share_save(_generate(spec, seed), share_dir)
        if barrier is not None:
            barrier()
        data = _share_load(share_dir)
    else:
        data = _generate(spec, seed)
    model = build_model(data, spec["L"], spec["w"], spec["S"], dof=spec["dof"], kind=kind, image_layers=spec["image_layers"])
    inputs = reference_inputs(data)
    if kind == "laue":
        col = lambda a, t: np.asarray(a).astype(t)[:, None]
        inputs = inputs + (col(data["wavelength"], np.float32), col(data["harmonic_id"], np.int64))
    spec["d"] = int(np.asarray(data["metadata"]).shape[1])
    spec["R"] = int(data["n_refl"])
    spec["M"] = int(data["n_images"])
    return model, inputs, data, spec
