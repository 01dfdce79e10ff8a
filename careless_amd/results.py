"""The output step right after the ELBO path: merged structure-factor amplitudes and per-observation predictions as plain arrays.

Array-level counterpart of `DataManager.get_results` / `get_predictions` (reference careless/io/manager.py:164-250, 89-161): the
reference wraps the same numbers into reciprocalspaceship DataSets and writes MTZ files (formatter / I-O scope, needs gemmi);
everything numerical happens here.  F / SigF are the truncated-normal moments of q, I = F^2 + SigF^2, SigI from the fourth moment
with the reference's I/SigI cap, N = observations per reflection; rows with N = 0 are flagged by `observed`.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from careless_amd.models.base import BaseModel


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def get_results(surrogate_posterior, inputs, output_parameters: bool = True, max_intensity_snr: float = 1e-5) -> Dict[str, np.ndarray]:
    """Reference manager.py:188-236 as a dict of per-reflection arrays:
    F, SigF, I, SigI, N, observed (N > 0) and, with `output_parameters`, the q parameters high / loc / low / scale."""
    q = surrogate_posterior
    if hasattr(q, "loc_raw"):                                # the truncated normal: one `cl_tn_moments` launch for all three
        from careless_amd.engine import tn_moments
        mom = tn_moments(q)                                  # (the reference takes <F^4> from scipy on the host, manager.py:192)
        F, SigF, f4 = _np(mom["mean"]), _np(mom["std"]), _np(mom["m4"])
    else:
        F = _np(q.mean()).astype(np.float32)
        SigF = _np(q.stddev()).astype(np.float32)
        f4 = np.asarray(q.moment_4(method="scipy"))          # <I^2> = <F^4>
    I = SigF * SigF + F * F
    ivar = np.square(I * max_intensity_snr)
    ivar = np.maximum(ivar, f4 - I * I)                      # var(I) = <F^4> - <I>^2, floored (manager.py:195-197)
    SigI = np.sqrt(ivar).astype(np.float32)
    refl_id = _np(BaseModel.get_refl_id(inputs)).reshape(-1).astype(np.int64)
    N = np.bincount(refl_id, minlength=len(F)).astype(np.float32)
    out = {"F": F, "SigF": SigF, "I": I.astype(np.float32), "SigI": SigI, "N": N, "observed": N > 0}
    if output_parameters:
        ones = np.ones(len(F), dtype=np.float32)
        for k in sorted(q.parameter_properties()):           # 'high', 'loc', 'low', 'scale' (manager.py:199-205)
            v = q.parameters[k]
            out[k] = (_np(v).reshape(-1) if not np.isscalar(v) else np.float32(v)) * ones
    return out


def get_predictions(model, inputs) -> Dict[str, np.ndarray]:
    """Per-observation posterior predictive moments (reference manager.py:89-161 -> variational.py:80-121, 47-78):
    Ipred, SigIpred, Scale, SigScale."""
    dist = model.scaling_model(inputs)                       # ONE forward pass of the scaler serves both pairs of moments
    iexp, isd = model.prediction_mean_stddev(inputs, scale_dist=dist)
    smean, sstd = model.scale_mean_stddev(inputs, scale_dist=dist)
    return {"Ipred": np.asarray(iexp, dtype=np.float32), "SigIpred": np.asarray(isd, dtype=np.float32),
            "Scale": np.asarray(smean, dtype=np.float32), "SigScale": np.asarray(sstd, dtype=np.float32)}
