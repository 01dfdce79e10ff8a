"""Learnable surrogate posterior over structure-factor amplitudes.

Mirror of `careless/models/merging/surrogate_posteriors.py:11-131` (reference): `TruncatedNormal` with
`from_loc_and_scale` (loc = Exp(raw), scale = Shift(eps)(Exp(raw))), `sample` clamped at `low`, `log_prob`, `mean`,
`stddev`, `moment_4`.  Sampling, log-prob and their gradients on the training path run in the HIP kernels
`cl_tn_forward` / `cl_tn_backward`; the moment accessors used by the output step (`mean`, `stddev`, `moment_4(method='tf')`) run
in `cl_tn_moments`.  `moment_4(method='scipy')` is scipy on the host, as in the reference.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def _t(x, device=None):
    if torch.is_tensor(x):
        return x.detach().to(dtype=torch.float32, device=device)
    return torch.as_tensor(np.asarray(x, dtype=np.float32), device=device)


def _ndtr(x):
    return torch.special.ndtr(x)


class SurrogatePosterior:
    """Base class for learnable variational distributions over structure factor amplitudes."""
    trainable = True

    def moment_4(self):
        raise NotImplementedError("The fourth moment of this distribution is not implemented yet.")


class TruncatedNormal(SurrogatePosterior):
    """q(F): one truncated normal per reflection (reference surrogate_posteriors.py:12-131).  `sample`, `mean`, `stddev`, `variance` and
    `moment_4` run on the GPU through the C-ABI (`cl_tn_forward`, `cl_tn_moments`: fp64 closed forms on the device) and return CUDA tensors
    without autograd history; without the library or a GPU they raise `CarelessHipError` (there is no CPU path: `engine.require_gpu`).  A
    host-side reader of a saved posterior (a pickle opened on a login node) takes the moments from the written MTZ columns, or from
    `scipy.stats.truncnorm((low - loc) / scale, (high - loc) / scale, loc, scale)` on `loc` / `scale`, which are plain tensors.
    `log_prob` is torch arithmetic on the parameters' device."""

    def __init__(self, loc_raw, scale_raw, low, high=1e10, scale_shift=1e-7):
        """Holds the *raw* trainable vectors a = log(loc), b = log(scale - scale_shift)."""
        self.loc_raw = _t(loc_raw)
        self.scale_raw = _t(scale_raw)
        low = _t(low)
        self.low = low.expand_as(self.loc_raw).contiguous() if low.dim() == 0 or low.numel() == 1 else low
        self.high = float(high)
        self.scale_shift = float(scale_shift)
        self.trainable = True

    @classmethod
    def from_loc_and_scale(cls, loc, scale, low=0.0, high=1e10, scale_shift=1e-7):
        """Instantiate a learnable distribution with the reference's bijectors (surrogate_posteriors.py:104-131)."""
        loc = np.asarray(loc.detach().cpu() if torch.is_tensor(loc) else loc, dtype=np.float64)
        scale = np.asarray(scale.detach().cpu() if torch.is_tensor(scale) else scale, dtype=np.float64)
        return cls(np.log(loc).astype(np.float32), np.log(scale - scale_shift).astype(np.float32), low, high, scale_shift)

    # -- parameters ---------------------------------------------------------------------------------------
    @property
    def loc(self):
        return torch.exp(self.loc_raw)

    @property
    def scale(self):
        return torch.exp(self.scale_raw) + self.scale_shift

    @property
    def parameters(self):
        return {"loc": self.loc, "scale": self.scale, "low": self.low, "high": self.high}

    def parameter_properties(self):
        return {"loc": None, "scale": None, "low": None, "high": None}

    @property
    def trainable_variables(self):
        return [self.loc_raw, self.scale_raw] if self.trainable else []

    def save_weights(self, path):
        torch.save({"loc_raw": self.loc_raw.detach().cpu(), "scale_raw": self.scale_raw.detach().cpu()}, path)

    def load_weights(self, path):
        st = torch.load(path)
        self.loc_raw.copy_(st["loc_raw"])
        self.scale_raw.copy_(st["scale_raw"])

    # -- distribution protocol ------------------------------------------------------------------------------
    def sample(self, sample_shape=(), seed=None):
        """max(low, TruncatedNormal sample) (surrogate_posteriors.py:50-53); runs `cl_tn_forward` on the GPU."""
        from careless_amd.engine import tn_sample
        n = 1 if sample_shape in ((), None) else int(sample_shape)
        z = tn_sample(self, n, seed=seed)
        return z[0] if sample_shape in ((), None) else z

    def _std_bounds(self):
        loc, scale = self.loc, self.scale
        return (self.low - loc) / scale, (self.high - loc) / scale

    def log_prob(self, z):
        z = _t(z, self.loc_raw.device)
        loc, scale = self.loc, self.scale
        a, b = self._std_bounds()
        zn = _ndtr(-a) - _ndtr(-b)
        y = (z - loc) / scale
        lp = -(0.5 * y * y + 0.5 * math.log(2 * math.pi) + torch.log(scale) + torch.log(zn))
        return torch.where((z < self.low) | (z > self.high), torch.full_like(lp, -math.inf), lp)

    def mean(self):
        """E[z] of the truncated normal (tfd.TruncatedNormal.mean behind surrogate_posteriors.py:23-24); `cl_tn_moments` on the GPU."""
        from careless_amd.engine import tn_moments
        return tn_moments(self, want=("mean",))["mean"]

    def stddev(self):
        """sqrt(Var[z]) (surrogate_posteriors.py:26-27); `cl_tn_moments` on the GPU."""
        from careless_amd.engine import tn_moments
        return tn_moments(self, want=("std",))["std"]

    def variance(self):
        sd = self.stddev()
        return sd * sd

    def moment_4(self, high=np.inf, method="scipy"):
        """Fourth raw moment (surrogate_posteriors.py:55-102).  'scipy' = scipy.stats.truncnorm.moment,
        'tf' = the closed form of `_tf_moment_4`, evaluated by `cl_tn_moments` on the GPU (fp64)."""
        if method == "scipy":
            from scipy.stats import truncnorm
            loc = self.loc.detach().cpu().numpy().astype(np.float64)
            scale = self.scale.detach().cpu().numpy().astype(np.float64)
            low = self.low.detach().cpu().numpy().astype(np.float64)
            hi = self.high if high is None else high
            a, b = (low - loc) / scale, (hi - loc) / scale
            return truncnorm.moment(4, a, b, loc, scale)
        if method == "tf":
            from careless_amd.engine import tn_moments
            hi = self.high if high is None else high
            return tn_moments(self, high_m4=hi, want=("m4",))["m4"].cpu().numpy()
        raise ValueError(f"Unknown method {method} for computing moment_4")
