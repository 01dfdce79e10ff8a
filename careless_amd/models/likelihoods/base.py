"""Protocol of a likelihood: called with `inputs`, it returns an object whose `log_prob(ipred)` scores predicted intensities of
shape (S, N) against the observations (reference careless/models/likelihoods/base.py).  In this package the objects are
descriptions (kind, degrees of freedom, error-model parameters); the arithmetic runs in the fused HIP kernel."""
from careless_amd.models.base import BaseModel


class Likelihood(BaseModel):
    kind = None          # "normal" | "studentt": what the engine dispatches on

    def call(self, inputs):
        raise NotImplementedError("Likelihoods must implement a call method that returns an object with a `log_prob` method.")
