"""Wilson prior on structure-factor amplitudes.

Mirror of `careless/models/priors/wilson.py:13-80` (reference).  On the hot path the prior is only a *description*
(centric flags, multiplicity, Sigma): its log-density and gradient are evaluated inside the HIP kernels
`cl_tn_forward` / `cl_tn_backward` (careless_amd/csrc/elbo_elem.hip).  `mean` / `stddev` seed the surrogate posterior
exactly as `DataManager.build_model` does (reference `careless/io/manager.py:432`), `log_prob` / `prob` are host-side
conveniences for users and tests.
"""
from __future__ import annotations

import math

import numpy as np

from careless_amd.models.priors.base import Prior


class WilsonPrior(Prior):
    """Wilson's priors on structure factor amplitudes."""

    def __init__(self, centric, epsilon, sigma=1.0):
        """
        centric : array, 1/True for centric reflections
        epsilon : array of multiplicities
        sigma   : float or array, the Wilson Sigma (mean intensity) per reflection
        """
        super().__init__()
        self.epsilon = np.array(epsilon, dtype=np.float32)
        self.centric = np.array(centric, dtype=bool)
        self.sigma = np.array(sigma, dtype=np.float32)

    # -- what the engine consumes -----------------------------------------------------------------
    @property
    def eps_sigma(self) -> np.ndarray:
        """multiplicity * Sigma per reflection, float32 (the `es` array of `cl_tn_args`)."""
        return (self.epsilon * self.sigma).astype(np.float32) * np.ones_like(self.epsilon)

    # -- reference protocol ------------------------------------------------------------------------
    def log_prob(self, x):
        """where(centric, HalfNormal(sqrt(eps Sigma)).log_prob(x), Weibull(2, sqrt(eps Sigma)).log_prob(x))
        (reference wilson.py:50-57)."""
        x = np.asarray(x, dtype=np.float64)
        es = self.eps_sigma.astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            lc = -0.5 * x * x / es + 0.5 * math.log(2.0 / math.pi) - 0.5 * np.log(es)
            la = math.log(2.0) + np.log(x) - np.log(es) - x * x / es
        return np.where(self.centric, lc, la).astype(np.float32)

    def prob(self, x):
        return np.exp(self.log_prob(x))

    def mean(self):
        """HalfNormal mean sigma sqrt(2/pi); Weibull(k=2) mean lambda Gamma(3/2)  (reference wilson.py:68-69)."""
        s = np.sqrt(self.eps_sigma.astype(np.float64))
        return np.where(self.centric, s * math.sqrt(2.0 / math.pi), s * math.gamma(1.5)).astype(np.float32)

    def stddev(self):
        """HalfNormal sigma sqrt(1-2/pi); Weibull(k=2) lambda sqrt(1-pi/4)  (reference wilson.py:71-72)."""
        s = np.sqrt(self.eps_sigma.astype(np.float64))
        return np.where(self.centric, s * math.sqrt(1.0 - 2.0 / math.pi),
                        s * math.sqrt(1.0 - math.pi / 4.0)).astype(np.float32)


class DoubleWilsonPrior(Prior):
    """Multivariate "double-Wilson" prior: every reflection of a non-root ASU is conditioned on its parent reflection
    (reference `careless/models/priors/wilson.py:82-175`, maths in reference `doc/double_wilson.md`).

    The reference constructor derives the parent lookup (`reflids`) from a `ReciprocalASUCollection` with
    reciprocalspaceship / gemmi, which is formatter territory (out of scope here); this class therefore takes the
    already-derived arrays.  `parents` / `r_values` keep the reference's meaning (one entry per ASU).

    reflids : (R,) int, id of the parent reflection, -1 where the parent is absent
    root    : (R,) bool, True for reflections of root ASUs (plain Wilson prior)
    asu_ids : (R,) int, ASU of every reflection
    """

    def __init__(self, centric, epsilon, reflids, root, asu_ids, r_values, parents=None, sigma=1.0, optimize_r=False):
        super().__init__()
        self.parents = parents
        self.optimize_r = bool(optimize_r)
        r0 = np.array(r_values, dtype=np.float32)
        for r in r0:
            if (r >= 1.0) or (r <= -1.0):                  # reference io/manager.py:415-419
                raise ValueError(f"Supplied --double-wilson-r value {r} outside of allowed range (-1, 1)")
        self._r_fixed = r0
        self.r_raw = None
        if self.optimize_r:
            # tfu.TransformedVariable(r, tfb.Sigmoid()) (reference wilson.py:105-110): the trainable value is logit(r)
            import torch
            with np.errstate(divide="ignore"):
                self.r_raw = torch.as_tensor(np.log(r0.astype(np.float64)) - np.log1p(-r0.astype(np.float64))).to(torch.float32)
        self.centric = np.array(centric, dtype=bool)
        self.multiplicity = np.array(epsilon, dtype=np.float32)
        self.asu_ids = np.array(asu_ids).reshape(-1).astype(np.int64)
        self.sigma = np.array(sigma, dtype=np.float32)
        self.reflids = np.array(reflids).reshape(-1).astype(np.int64)
        self.absent = self.reflids == -1
        self.root = np.array(root, dtype=bool)
        self.wilson_prior = WilsonPrior(self.centric, self.multiplicity, sigma)

    @property
    def eps_sigma(self):
        return self.wilson_prior.eps_sigma

    @property
    def r(self) -> np.ndarray:
        """Current correlation per ASU (sigmoid of the trainable value when --optimize-double-wilson-r is on)."""
        if self.r_raw is None:
            return self._r_fixed
        import torch
        return torch.sigmoid(self.r_raw.detach().float()).cpu().numpy()

    @property
    def trainable_variables(self):
        return [self.r_raw] if self.r_raw is not None else []

    @property
    def r_per_reflection(self) -> np.ndarray:
        return self.r[self.asu_ids].astype(np.float32)

    def mean(self):
        return self.wilson_prior.mean()

    def stddev(self):
        return self.wilson_prior.stddev()

    def log_prob(self, z):
        """Host-side evaluation (float64 numpy/scipy) of reference wilson.py:146-175, for users and tests."""
        from scipy import special
        z = np.asarray(z, dtype=np.float64)
        r = self.r_per_reflection.astype(np.float64)
        mask = self.reflids >= 0
        zp = np.where(mask, z[..., np.where(mask, self.reflids, 0)], 0.0)
        loc = np.where(self.absent, 0.0, zp * r)
        es = self.eps_sigma.astype(np.float64)
        scale = np.where(self.centric, np.sqrt(es * (1 - r * r)), np.sqrt(0.5 * es * (1 - r * r)))
        with np.errstate(divide="ignore", invalid="ignore"):
            arg = z * loc / scale ** 2
            rice = np.log(z) - 2 * np.log(scale) - (z * z + loc * loc) / (2 * scale ** 2) + np.log(special.i0e(arg)) + np.abs(arg)
            fn = np.logaddexp(-0.5 * ((z - loc) / scale) ** 2, -0.5 * ((-z - loc) / scale) ** 2) - 0.5 * math.log(2 * math.pi) - np.log(scale)
        p_dw = np.where(self.centric, fn, rice)
        return np.where(self.root, self.wilson_prior.log_prob(z), p_dw).astype(np.float32)
