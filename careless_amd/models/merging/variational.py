"""Variational merging model: the training driver of careless on MI355X.

Mirror of `careless/models/merging/variational.py:11-275` (reference): same constructor, `train_model` signature and
returned history (`{"Grad Norm", "loss", "F KLDiv", "NLL"}` lists, early stop on a non-finite gradient norm),
`__call__(inputs) -> ipred (S, N)`, `scale_mean_stddev`, `prediction_mean_stddev`.  The per-step arithmetic is the HIP
engine (careless_amd/engine.py -> libcareless_hip.so); there is no eager / CPU path.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from careless_amd.models.base import BaseModel
from careless_amd.optimizers import Adam


class VariationalMergingModel(BaseModel):
    """Merge data with a posterior parameterized by a surrogate distribution."""

    def __init__(self, surrogate_posterior, prior, likelihood, scaling_model, mc_sample_size=1, kl_weight=None,
                 scale_kl_weight=None, scale_prior=None):
        super().__init__()
        self.prior = prior
        self.surrogate_posterior = surrogate_posterior
        self.likelihood = likelihood
        self.scaling_model = scaling_model
        self.mc_sample_size = mc_sample_size
        self.kl_weight = kl_weight
        self.scale_kl_weight = scale_kl_weight
        self.scale_prior = scale_prior
        self.optimizer = Adam()
        self.seed = 1234              # careless/args/tf_options.py:50-54
        self._engine = None
        self._engine_inputs = None
        self._dist = None             # (rank, world, process_group) for data-parallel runs

    # -- keras-like surface -------------------------------------------------------------------------------
    def compile(self, optimizer=None, run_eagerly=None, **kwargs):
        """`model.compile(opt, run_eagerly=...)` (reference io/manager.py:503-506); run_eagerly has no meaning here."""
        if optimizer is None or optimizer == "Adam":
            optimizer = Adam()
        self.optimizer = optimizer
        return self

    def set_data_parallel(self, rank: int, world: int, process_group=None):
        """Shard the observations of every later `train_model` call over `world` ranks (one process per GPU)."""
        self._dist = (int(rank), int(world), process_group)
        self._engine = None

    @property
    def trainable_variables(self):
        return self.surrogate_posterior.trainable_variables + self.scaling_model.trainable_variables

    # -- engine management -------------------------------------------------------------------------------
    def engine(self, inputs):
        from careless_amd.engine import ElboEngine, make_shard
        if self._engine is None or self._engine_inputs is not inputs:
            shard = None
            pg = None
            if self._dist is not None:
                n = int(np.asarray(BaseModel.get_refl_id(inputs).shape)[0])
                rank, world, pg = self._dist
                shard = make_shard(n, int(self.surrogate_posterior.loc_raw.numel()), rank, world)
            self._engine = ElboEngine(self, inputs, seed=self.seed, shard=shard, process_group=pg,
                                      grid=getattr(self, "kernel_grid", None))   # None: one persistent workgroup per CU
            self._engine_inputs = inputs
        else:
            self._engine.refresh_config()
        return self._engine

    # -- forward ------------------------------------------------------------------------------------------
    def call(self, inputs, u_f=None, eta=None):
        """Predictions `ipred` (S, N) for one draw of q(F) and q(Sigma) (reference variational.py:141-183)."""
        eng = self.engine(inputs)
        du, de = eng._noise_to_device(u_f, eta)
        ipred = torch.empty(eng.N * eng.S, dtype=torch.float32, device=eng.device)
        eng.forward_backward(eng.t, du, de, ipred_out=ipred)
        perm = getattr(eng.obs, "perm", None)
        if perm is not None:                    # rows stored in image order (per-image layers on the layer-by-layer path): back to the caller's
            out = torch.empty_like(ipred).view(eng.N, eng.S)
            out[torch.as_tensor(perm, device=eng.device)] = ipred.view(eng.N, eng.S)
            return out.t()
        return ipred.view(eng.N, eng.S).t()

    # -- training ------------------------------------------------------------------------------------------
    def train_model(self, data, steps, message=None, format_string="{:0.2e}", validation_data=None,
                    validation_frequency=10, progress=True, use_custom_train_step=True, jit_compile=None,
                    reduce_retracing=False, noise=None, validation_noise=None):
        """Full-batch ELBO optimisation for `steps` iterations (reference variational.py:226-275).

        `noise`: optional callable step -> (u_f (S,R), eta (S,N)) injecting the Monte-Carlo noise (parity tests);
        by default the kernels draw it with the counter-based generator keyed by (seed, iteration).
        `validation_noise`: the same for the validation pass, eta of shape (S, N_validation)."""
        eng = self.engine(data)
        eng.alloc_history(steps)
        val_obs, val_scale, nll_val, val_hist = None, 1.0, float("nan"), []
        if validation_data is not None:                       # reference variational.py:248-249, 257-260
            val_obs = eng.make_obs(validation_data)
            val_scale = eng.N_total / val_obs.N_total
        bar = None
        if progress:
            try:
                from tqdm import trange
                bar = trange(steps, desc=message)
            except Exception:
                bar = None
        check_every = 50
        done = 0
        for i in range(steps):
            u_f = eta = None
            if noise is not None:
                u_f, eta = noise(i)
                u_f, eta = eng._noise_to_device(u_f, eta)
            eng.train_step(i, u_f, eta)
            if val_obs is not None:
                if i % validation_frequency == 0:             # the stale value is re-logged in between (:257-260)
                    vu, ve = validation_noise(i) if validation_noise is not None else (None, None)
                    nll_val = val_scale * eng.evaluate_nll(val_obs, (eng.t & 0x3FFFFFFF) | 0x40000000, vu, ve)
                val_hist.append(nll_val)
            done = i + 1
            if bar is not None:
                bar.update(1)
            if (i + 1) % check_every == 0 and int(eng.stop_flag.item()) != 0:
                break
        if bar is not None:
            bar.close()
        eng.sync_owned()              # reflection-owner data parallelism: every rank gets the other owners' q(F) parameters
        history = eng.read_history(done)
        if val_obs is not None:
            history["NLL_val"] = val_hist[: len(history["loss"])]
        if len(history["loss"]) < done or (len(history["Grad Norm"]) and not np.isfinite(history["Grad Norm"][-1])):
            print("Encountered numerical issues, terminating optimization early!")
        return history

    # -- output-step helpers ---------------------------------------------------------------------------------
    def _convolved(self, inputs):
        """The likelihood's `convolve` for Laue data, None for monochromatic data (reference variational.py:70-76, 111-119)."""
        from careless_amd.models.likelihoods.laue import LaueBase
        if isinstance(self.likelihood, LaueBase):
            return self.likelihood(inputs).convolve
        return None

    def scale_mean_stddev(self, inputs, scale_dist=None):
        """Moments of the posterior scale of every observation (reference variational.py:47-78).  Laue data: the moments of the
        rows of a harmonic group are summed into the group's slot (means add, variances add), slots without rows stay 0."""
        dist = self.scaling_model(inputs) if scale_dist is None else scale_dist
        mean, stddev = dist.mean().cpu().numpy(), dist.stddev().cpu().numpy()
        convolve = self._convolved(inputs)
        if convolve is not None:
            mean = convolve(mean)
            stddev = np.sqrt(convolve(stddev * stddev))
        return mean, stddev

    def prediction_mean_stddev(self, inputs, scale_dist=None):
        """E[I] and sd[I] of every observation under the current model (reference variational.py:80-121); for Laue data
        per harmonic slot: `iexp` and `ivar` of the member rows are summed before the square root (:113-119).  `scale_dist`: the
        scaler's output on `inputs` when the caller already has it (the output step asks for the scale moments too)."""
        rid = self.get_refl_id(inputs)
        refl_id = torch.as_tensor(np.asarray(rid.cpu() if torch.is_tensor(rid) else rid).reshape(-1).astype(np.int64))
        q = self.surrogate_posterior
        dist = self.scaling_model(inputs) if scale_dist is None else scale_dist
        if hasattr(q, "loc_raw") and dist.mean().is_cuda:
            # truncated normal (what the command line builds): the posterior's moments from one `cl_tn_moments` launch, the rows' from one
            # `cl_predict_moments` launch -- nothing per observation is computed on the host
            from careless_amd.engine import predict_moments, tn_moments
            iexp, ivar = predict_moments(dist.mean(), dist.stddev(), refl_id, tn_moments(q))
            convolve = self._convolved(inputs)
            if convolve is not None:
                iexp, ivar = convolve(iexp), convolve(ivar)
            return iexp.astype(np.float32), np.sqrt(ivar).astype(np.float32)
        smean, sstd = dist.mean().double().cpu(), dist.stddev().double().cpu()
        if hasattr(q, "loc_raw"):                           # truncated normal: mean, stddev and <F^4> from one `cl_tn_moments` launch
            from careless_amd.engine import tn_moments
            mom = tn_moments(q)
            f2 = (mom["mean"].double() ** 2 + mom["std"].double() ** 2).cpu()
            f4 = mom["m4"].cpu()
        else:
            f2 = (q.mean().double() ** 2 + q.stddev().double() ** 2).cpu()
            f4 = torch.as_tensor(np.asarray(q.moment_4(method="scipy"), dtype=np.float64))
        iexp = smean * f2[refl_id]
        s2 = smean ** 2 + sstd ** 2
        ivar = f4[refl_id] * s2 - iexp * iexp              # var(I) = <F^4><Sigma^2> - <I>^2
        iexp, ivar = iexp.numpy(), ivar.numpy()
        convolve = self._convolved(inputs)
        if convolve is not None:
            iexp, ivar = convolve(iexp), convolve(ivar)
        return iexp.astype(np.float32), np.sqrt(ivar).astype(np.float32)
