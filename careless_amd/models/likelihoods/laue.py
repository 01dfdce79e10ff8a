"""Laue (polychromatic) likelihoods.

Mirror of `careless/models/likelihoods/laue.py:9-100` (reference): `likelihood(inputs)` returns a `ConvolvedLikelihood` whose
`convolve(value)` sums the predictions of rows that share a `harmonic_id` and whose `log_prob(value)` evaluates the base
distribution on the convolved predictions over ALL N slots.  These host objects serve users and tests; on the training path the
engine reads `kind` / `dof` and runs the harmonic deconvolution in HIP (careless_amd/csrc/elbo_laue.hip).
"""
from __future__ import annotations

import numpy as np

from careless_amd.models.likelihoods.base import Likelihood
from careless_amd.models.likelihoods.mono import _BoundLocationScale, _squeeze


class ConvolvedLikelihood:
    """Convolved log probability object for Laue data (reference laue.py:9-34)."""

    def __init__(self, distribution, harmonic_id):
        self.harmonic_id = np.asarray(_squeeze(harmonic_id)).reshape(-1).astype(np.int64)
        self.distribution = distribution

    def convolve(self, value):
        """value: (n_predictions,) or (b, n_predictions); duplicates sum, untouched slots stay 0 (scatter_nd)."""
        value = np.asarray(value.detach().cpu() if hasattr(value, "detach") else value)
        rows = value.reshape(-1, value.shape[-1])
        out = np.stack([np.bincount(self.harmonic_id, weights=r, minlength=r.shape[0]) for r in rows])
        return out.reshape(value.shape).astype(value.dtype if value.dtype.kind == "f" else np.float64)

    def mean(self, *args, **kwargs):
        return self.distribution.mean(*args, **kwargs)

    def stddev(self, *args, **kwargs):
        return self.distribution.stddev(*args, **kwargs)

    def log_prob(self, value):
        return self.distribution.log_prob(self.convolve(value))


class LaueBase(Likelihood):
    kind = None
    dof = None

    def dist(self, inputs):
        raise NotImplementedError("Extensions of this class must implement dist(inputs)")

    def call(self, inputs):
        return ConvolvedLikelihood(self.dist(inputs), self.get_harmonic_id(inputs))


class NormalLikelihood(LaueBase):
    kind = "normal"

    def dist(self, inputs):
        return _BoundLocationScale("normal", _squeeze(self.get_intensities(inputs)), _squeeze(self.get_uncertainties(inputs)))


class StudentTLikelihood(LaueBase):
    kind = "studentt"

    def __init__(self, dof):
        super().__init__()
        self.dof = dof

    def dist(self, inputs):
        return _BoundLocationScale("studentt", _squeeze(self.get_intensities(inputs)),
                                   _squeeze(self.get_uncertainties(inputs)), dof=self.dof)


class _Ev11Laue(LaueBase):
    """Laue wrapper of the mono Evans-2011 likelihoods (reference laue.py:49-65): the error model sees the convolved prediction."""
    ev11 = True
    _mono_cls = None

    def __init__(self, *args):
        super().__init__()
        from careless_amd.models.likelihoods import mono as _mono
        self.mono = getattr(_mono, self._mono_cls)(*args)
        self.kind, self.dof = self.mono.kind, self.mono.dof

    @property
    def raw(self):
        return self.mono.raw

    @raw.setter
    def raw(self, v):
        self.mono.raw = v

    @property
    def trainable_variables(self):
        return self.mono.trainable_variables

    def dist(self, inputs):
        return self.mono(inputs)


class NormalEv11Likelihood(_Ev11Laue):
    _mono_cls = "NormalEv11Likelihood"


class StudentTEv11Likelihood(_Ev11Laue):
    _mono_cls = "StudentTEv11Likelihood"

    def __init__(self, dof):
        super().__init__(dof)
