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
