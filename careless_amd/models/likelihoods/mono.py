"""Monochromatic likelihoods.

Mirror of `careless/models/likelihoods/mono.py:10-37` (reference).  `likelihood(inputs)` returns a small object
bound to (Iobs, SigIobs) with `.log_prob(ipred)`, like the tfd distribution the reference returns.  On the hot path
the engine reads `kind` / `dof` from the likelihood and evaluates log-prob and its derivative inside the fused HIP
kernel `cl_elbo_mono_fwd_bwd` (careless_amd/csrc/elbo_mlp.hip).
"""
from __future__ import annotations

import math

import numpy as np

from careless_amd.models.likelihoods.base import Likelihood


def _squeeze(x):
    x = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
    return np.squeeze(x)


class _BoundLocationScale:
    def __init__(self, kind, loc, scale, dof=None):
        self.kind, self.loc, self.scale, self.dof = kind, loc, scale, dof

    def log_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        y = (x - self.loc) / self.scale
        if self.kind == "normal":
            return (-0.5 * y * y - 0.5 * math.log(2 * math.pi) - np.log(self.scale)).astype(np.float32)
        nu = float(self.dof)
        return (-0.5 * (nu + 1.0) * np.log1p(y * y / nu) - np.log(np.abs(self.scale)) - 0.5 * math.log(nu)
                - 0.5 * math.log(math.pi) - math.lgamma(0.5 * nu) + math.lgamma(0.5 * (nu + 1.0))).astype(np.float32)

    def mean(self):
        return self.loc

    def stddev(self):
        return self.scale


class LocationScaleLikelihood(Likelihood):
    kind = None
    dof = None

    def get_loc_and_scale(self, inputs):
        return _squeeze(self.get_intensities(inputs)), _squeeze(self.get_uncertainties(inputs))


class NormalLikelihood(LocationScaleLikelihood):
    """Normal(Iobs, SigIobs)  (reference mono.py:16-18)."""
    kind = "normal"

    def call(self, inputs):
        return _BoundLocationScale("normal", *self.get_loc_and_scale(inputs))


class StudentTLikelihood(LocationScaleLikelihood):
    """StudentT(dof, Iobs, SigIobs)  (reference mono.py:25-37)."""
    kind = "studentt"

    def __init__(self, dof):
        super().__init__()
        self.dof = dof

    def call(self, inputs):
        return _BoundLocationScale("studentt", *self.get_loc_and_scale(inputs), dof=self.dof)


_SOFTPLUS_INV_ONE = float(np.log(np.e - 1.0))      # softplus(raw) == 1


class Ev11Likelihood(LocationScaleLikelihood):
    """Evans-2011 error model with three learnable scalars Sdfac, Sdadd, SdB (each Softplus-transformed, initial value 1):
    sigma_c = Sdfac * sqrt(SigIobs^2 + SdB * softplus(ipred) + Sdadd * softplus(ipred)^2)  (reference mono.py:39-59)."""
    ev11 = True

    def __init__(self, *args, **kwargs):
        super().__init__()
        import torch
        # raw (pre-softplus) values in the reference's variable order: Sdfac, Sdadd, SdB
        self.raw = torch.full((3,), _SOFTPLUS_INV_ONE, dtype=torch.float32)
        self.loc = None
        self.scale = None

    @property
    def Sdfac(self):
        return float(np.log1p(np.exp(float(self.raw[0]))))

    @property
    def Sdadd(self):
        return float(np.log1p(np.exp(float(self.raw[1]))))

    @property
    def SdB(self):
        return float(np.log1p(np.exp(float(self.raw[2]))))

    @property
    def trainable_variables(self):
        return [self.raw]

    def call(self, inputs):
        self.loc, self.scale = self.get_loc_and_scale(inputs)
        return self

    def corrected_sigiobs(self, ipred):
        ipred = np.asarray(ipred, dtype=np.float64)
        sp = np.logaddexp(0.0, ipred)
        return self.Sdfac * np.sqrt(np.square(self.scale) + self.SdB * sp + self.Sdadd * np.square(sp))

    def log_prob(self, ipred):
        return _BoundLocationScale(self.kind, self.loc, self.corrected_sigiobs(ipred), dof=self.dof).log_prob(ipred)


class NormalEv11Likelihood(Ev11Likelihood):
    kind = "normal"


class StudentTEv11Likelihood(Ev11Likelihood):
    kind = "studentt"

    def __init__(self, dof, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.dof = dof
