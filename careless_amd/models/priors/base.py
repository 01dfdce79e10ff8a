from careless_amd.models.base import BaseModel


class Prior(BaseModel):
    """Base class for prior distributions on merged normalized structure factor amplitudes
    (reference `careless/models/priors/base.py`)."""

    def log_prob(self, x):
        raise NotImplementedError("No log_prob method defined. All Priors must implement a log_prob method")
