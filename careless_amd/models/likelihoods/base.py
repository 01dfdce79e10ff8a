from careless_amd.models.base import BaseModel


class Likelihood(BaseModel):
    """Reference `careless/models/likelihoods/base.py`."""

    def call(self, inputs):
        raise NotImplementedError(
            "Likelihoods must implement a call method that returns an object with a `log_prob` method.")
