"""Protocol of a prior over merged structure-factor amplitudes: `log_prob(z)` for samples z of shape (S, R) or (R,), plus `mean()`
and `stddev()` used to initialise the surrogate posterior (reference careless/models/priors/base.py, io/manager.py:432)."""
from careless_amd.models.base import BaseModel


class Prior(BaseModel):
    def log_prob(self, x):
        raise NotImplementedError("No log_prob method defined. All Priors must implement a log_prob method")

    def mean(self):
        raise NotImplementedError(f"{type(self).__name__} does not define mean()")

    def stddev(self):
        raise NotImplementedError(f"{type(self).__name__} does not define stddev()")
