"""CPU oracle for the careless per-step Monte-Carlo ELBO (TEST INFRASTRUCTURE ONLY).

This file is a CPU restatement, in plain PyTorch-CPU (fp64 by default) + scipy, of the algorithm that
`rs-station/careless` v0.5.4 executes for one ELBO training step.  It is the *checker* for the HIP
kernels under `careless_amd/csrc/`; it is never imported by the product package.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.

PARITY STATUS -- "parity unpinned by the reference":
  the reference's arithmetic lives in un-vendored third-party packages (tensorflow==2.18.0,
  tensorflow-probability==0.25, tf_keras; reference `pyproject.toml:14-19`) that are absent from the
  build container, and the reference's own tests hold no golden ELBO / gradient / trajectory vectors.
  What pins this oracle instead (see tests/test_oracle_kat.py):
    * the reference's closed-form known-answer tests restated with scipy as the independent check
      (`tests/models/priors/test_wilson.py:13-29`, `tests/models/merging/test_truncated_normal.py:29-42`,
      `tests/models/likelihoods/test_mono.py:12-51`, `tests/models/likelihoods/test_laue.py:11-36`);
    * scipy.stats closed forms for every density used (truncnorm, norm, t, halfnorm, weibull_min,
      rice, foldnorm);
    * a finite-difference check of the truncated-normal pathwise gradient;
    * committed golden vectors produced by this file (tests/golden/, generator committed beside them).

Every function cites the reference file:line it follows (paths relative to the reference checkout).
Statements about TF / TFP internals are recalled from the pinned versions and marked [3P].
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# float32 constants used by TFP's truncated-normal sample gradient [3P]
TINY_F32 = float(np.finfo(np.float32).tiny)
EPS_F32 = float(np.finfo(np.float32).eps)
LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------------------
@dataclass
class ElboConfig:
    """Static configuration of one ELBO graph (what `DataManager.build_model` wires up,
    reference `careless/io/manager.py:380-507`)."""
    mc_samples: int = 1                 # --mc-samples (args/common.py:11-15)
    likelihood: str = "normal"          # "normal" | "studentt"  (manager.py:438-443)
    dof: Optional[float] = None         # --studentt-likelihood-dof
    scale_bijector: str = "exp"         # --scale-bijector (args/scaling.py:47-52)
    epsilon: float = 1e-7               # --epsilon: q scale shift AND scaler sigma shift (manager.py:436,453,460)
    scale_shift: float = 0.0            # istd: tfb.Shift(std(Iobs)) for the softplus bijector (nn.py:84-87, manager.py:457)
    use_image_scales: bool = True       # HybridImageScaler is the CLI default (manager.py:484-489)
    kl_weight: Optional[float] = None   # --kl-weight (variational.py:172-177)
    leakiness: float = 0.01             # nn.py:31
    high: float = 1e10                  # surrogate_posteriors.py:105
    laue: bool = False                  # len(inputs) >= 8 (models/base.py:39-47)
    prior: str = "wilson"               # "wilson" | "double_wilson"
    ev11: bool = False                  # --refine-uncertainties: Evans-2011 error model (likelihoods/mono.py:39-73)
    optimize_dw_r: bool = False         # --optimize-double-wilson-r (priors/wilson.py:105-110)
    image_layers: int = 0               # --image-layers: NeuralImageScaler (scaling/image.py:98-125); replaces the image scales
    # Adam (manager.py:494-501; args/optimizer.py)
    learning_rate: float = 1e-3
    beta_1: float = 0.9
    beta_2: float = 0.99
    adam_epsilon: float = 1e-7          # tf_keras Adam default [3P]
    clipnorm: Optional[float] = None
    clipvalue: Optional[float] = None
    global_clipnorm: Optional[float] = None


# --------------------------------------------------------------------------------------------------
# special functions
# --------------------------------------------------------------------------------------------------
def ndtr(x: torch.Tensor) -> torch.Tensor:
    return torch.special.ndtr(x)


def log_ndtr(x: torch.Tensor) -> torch.Tensor:
    return torch.special.log_ndtr(x)


def ndtri(p: torch.Tensor) -> torch.Tensor:
    return torch.special.ndtri(p)


def log_i0e_plus_abs(x: torch.Tensor) -> torch.Tensor:
    """log(I0(x)) = log(i0e(x)) + |x|  (reference `careless/utils/distributions.py:260-261`)."""
    return torch.log(torch.special.i0e(x)) + torch.abs(x)


# --------------------------------------------------------------------------------------------------
# surrogate posterior q(F): truncated normal  (careless/models/merging/surrogate_posteriors.py:45-131)
# --------------------------------------------------------------------------------------------------
def tn_loc_scale(q_loc_raw: torch.Tensor, q_scale_raw: torch.Tensor, epsilon: float):
    """`from_loc_and_scale` (surrogate_posteriors.py:104-131): loc = Exp(raw), scale = Shift(eps)(Exp(raw))."""
    return torch.exp(q_loc_raw), torch.exp(q_scale_raw) + epsilon


def tn_raw_from_loc_scale(loc, scale, epsilon: float):
    """Inverse of the bijectors: what TransformedVariable stores as its trainable pretransformed value [3P]."""
    loc = np.asarray(loc, dtype=np.float64)
    scale = np.asarray(scale, dtype=np.float64)
    return np.log(loc), np.log(scale - epsilon)


def tn_log_normalizer(alpha: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """log(Phi(beta) - Phi(alpha)), evaluated as Phi(-alpha) - Phi(-beta) when alpha > 0 for accuracy."""
    z = torch.where(alpha > 0, ndtr(-alpha) - ndtr(-beta), ndtr(beta) - ndtr(alpha))
    return torch.log(z)


class _TNStdSample(torch.autograd.Function):
    """Standardised truncated-normal sample e ~ N(0,1) restricted to [alpha, beta], drawn by inverse CDF
    from an injected uniform `u`, with the pathwise gradient TFP attaches to its sampler [3P]
    (`tfd.TruncatedNormal._sample_n` -> `_std_samples_with_gradients`): with
        cdf = clip((Phi(e) - Phi(alpha)) / (Phi(beta) - Phi(alpha)), tiny_f32, 1 - eps_f32)
        dl  = exp(0.5 (e^2 - alpha^2) + log1p(-cdf)),   du = exp(0.5 (e^2 - beta^2) + log cdf)
    the sample gradients are de/dalpha = dl and de/dbeta = du.
    The reference draws e with TF's rejection sampler; the *distribution* is the same and parity is defined on
    injected noise, so both the oracle and the HIP kernel use the inverse-CDF map of the same `u`."""

    @staticmethod
    def forward(ctx, alpha, beta, u):
        lo = ndtr(alpha)
        z = torch.where(alpha > 0, ndtr(-alpha) - ndtr(-beta), ndtr(beta) - lo)
        p = lo + u * z
        q = ndtr(-beta) + (1.0 - u) * z
        e = torch.where(p < 0.5, ndtri(p), -ndtri(q))
        # guard the interval against round-off
        e = torch.minimum(torch.maximum(e, alpha), beta)
        ctx.save_for_backward(e, alpha, beta)
        return e

    @staticmethod
    def backward(ctx, g):
        e, alpha, beta = ctx.saved_tensors
        z = torch.where(alpha > 0, ndtr(-alpha) - ndtr(-beta), ndtr(beta) - ndtr(alpha))
        cdf = (ndtr(e) - ndtr(alpha)) / z
        cdf = torch.clamp(cdf, TINY_F32, 1.0 - EPS_F32)
        du = torch.exp(0.5 * (e * e - beta * beta) + torch.log(cdf))
        dl = torch.exp(0.5 * (e * e - alpha * alpha) + torch.log1p(-cdf))
        ga = (g * dl)
        gb = (g * du)
        # reduce broadcast (S,R) -> (R,)
        while ga.dim() > alpha.dim():
            ga = ga.sum(0)
            gb = gb.sum(0)
        return ga, gb, None


def tn_sample(loc, scale, low, high, u):
    """`TruncatedNormal.sample` (surrogate_posteriors.py:50-53): tf.maximum(low, distribution.sample()).
    u: (S, R) uniforms in (0,1).  Returns z (S, R)."""
    alpha = (low - loc) / scale
    beta = (high - loc) / scale
    e = _TNStdSample.apply(alpha, beta, u)
    s = e * scale + loc
    # tf.maximum(low, s): gradient goes to `low` (a constant) on ties / when s < low
    return torch.where(s > low, s, low.expand_as(s) if torch.is_tensor(low) else torch.full_like(s, low))


def tn_log_prob(z, loc, scale, low, high):
    """`tfd.TruncatedNormal.log_prob` [3P] (called at variational.py:128 via surrogate_posteriors.py:20-21):
    -(0.5 ((z-loc)/scale)^2 + 0.5 log 2pi + log scale + log(Phi(beta)-Phi(alpha))), -inf outside [low, high]."""
    alpha = (low - loc) / scale
    beta = (high - loc) / scale
    y = (z - loc) / scale
    lp = -(0.5 * y * y + 0.5 * LOG_2PI + torch.log(scale) + tn_log_normalizer(alpha, beta))
    bad = (z < low) | (z > high)
    return torch.where(bad, torch.full_like(lp, -math.inf), lp)


def tn_mean(loc, scale, low, high):
    """`tfd.TruncatedNormal.mean` [3P] (used by manager.py:188, variational.py:100)."""
    alpha = (low - loc) / scale
    beta = (high - loc) / scale
    zn = torch.exp(tn_log_normalizer(alpha, beta))
    pa = torch.exp(-0.5 * alpha * alpha) / math.sqrt(2 * math.pi)
    pb = torch.exp(-0.5 * beta * beta) / math.sqrt(2 * math.pi)
    return loc + scale * (pa - pb) / zn


def tn_variance(loc, scale, low, high):
    """`tfd.TruncatedNormal.variance` [3P]."""
    alpha = (low - loc) / scale
    beta = (high - loc) / scale
    zn = torch.exp(tn_log_normalizer(alpha, beta))
    pa = torch.exp(-0.5 * alpha * alpha) / math.sqrt(2 * math.pi)
    pb = torch.exp(-0.5 * beta * beta) / math.sqrt(2 * math.pi)
    # beta * pb -> 0 for the huge `high` used by careless
    bpb = torch.where(torch.isfinite(beta) & (pb > 0), beta * pb, torch.zeros_like(pb))
    r = (pa - pb) / zn
    return scale * scale * (1.0 + (alpha * pa - bpb) / zn - r * r)


def tn_moment_4(loc, scale, low):
    """`TruncatedNormal._tf_moment_4` with high=inf (surrogate_posteriors.py:55-73): closed form of <F^4>."""
    a = low
    mu, sigma = loc, scale
    z_a = (a - mu) / sigma
    pdf_a = torch.exp(-0.5 * z_a * z_a) / math.sqrt(2 * math.pi)
    aterm = (a * a * a + a * a * mu + a * mu * mu + sigma * sigma * (3 * a + 5 * mu) + mu * mu * mu) * pdf_a
    num = -aterm
    den = 1.0 - ndtr(z_a)
    return mu ** 4 + 6 * mu * mu * sigma * sigma + 3 * sigma ** 4 - sigma * num / den


# --------------------------------------------------------------------------------------------------
# priors  (careless/models/priors/wilson.py)
# --------------------------------------------------------------------------------------------------
def wilson_log_prob(z, centric, multiplicity, sigma):
    """`WilsonPrior.log_prob` (wilson.py:50-57): where(centric, HalfNormal(sqrt(eps*Sigma)), Weibull(2, sqrt(eps*Sigma))).
    HalfNormal/Weibull densities per TFP [3P]; closed forms pinned by the reference's tests/models/priors/test_wilson.py:13-29."""
    es = multiplicity * sigma
    lp_c = -0.5 * z * z / es + 0.5 * math.log(2.0 / math.pi) - 0.5 * torch.log(es)
    lp_a = math.log(2.0) + torch.log(z) - torch.log(es) - z * z / es
    return torch.where(centric, lp_c, lp_a)


def wilson_mean(centric, multiplicity, sigma):
    """`WilsonPrior.mean` (wilson.py:68-69): HalfNormal mean sigma sqrt(2/pi); Weibull(k=2) mean lambda Gamma(1.5)."""
    s = np.sqrt(np.asarray(multiplicity, dtype=np.float64) * sigma)
    return np.where(centric, s * np.sqrt(2.0 / np.pi), s * math.gamma(1.5))


def wilson_stddev(centric, multiplicity, sigma):
    """`WilsonPrior.stddev` (wilson.py:71-72): HalfNormal sigma sqrt(1-2/pi); Weibull(k=2) lambda sqrt(1 - pi/4)."""
    s = np.sqrt(np.asarray(multiplicity, dtype=np.float64) * sigma)
    return np.where(centric, s * np.sqrt(1.0 - 2.0 / np.pi), s * np.sqrt(1.0 - np.pi / 4.0))


def rice_log_prob(x, nu, sigma):
    """`Rice.log_prob` (careless/utils/distributions.py:278-283)."""
    return (torch.log(x) - 2.0 * torch.log(sigma) - (x * x + nu * nu) / (2.0 * sigma * sigma)
            + log_i0e_plus_abs(x * nu / (sigma * sigma)))


def folded_normal_log_prob(x, loc, scale):
    """`FoldedNormal.log_prob` (distributions.py:333-335): TransformedDistribution(Normal, AbsoluteValue) [3P]:
    log(N(x; loc, scale) + N(-x; loc, scale)); NaN for x < 0."""
    la = -0.5 * ((x - loc) / scale) ** 2
    lb = -0.5 * ((-x - loc) / scale) ** 2
    lp = torch.logaddexp(la, lb) - 0.5 * LOG_2PI - torch.log(scale)
    return torch.where(x < 0, torch.full_like(lp, math.nan), lp)


def double_wilson_log_prob(z, centric, multiplicity, sigma, parent_ids, root, asu_ids, r):
    """`DoubleWilsonPrior.log_prob` (wilson.py:146-175).
    z: (S,R); parent_ids: (R,) int with -1 for an absent parent; root: (R,) bool; asu_ids: (R,) int; r: (n_asu,)."""
    rr = r[asu_ids]
    mask = parent_ids >= 0
    san = torch.where(mask, parent_ids, torch.zeros_like(parent_ids))
    z_parent = torch.where(mask[None, :], z[..., san], torch.zeros_like(z))
    loc = torch.where(~mask, torch.zeros_like(z_parent), z_parent * rr)
    r2 = rr * rr
    scale = torch.where(centric, torch.sqrt(multiplicity * sigma * (1.0 - r2)),
                        torch.sqrt(0.5 * multiplicity * sigma * (1.0 - r2)))
    p_dw = torch.where(centric, folded_normal_log_prob(z, loc, scale), rice_log_prob(z, loc, scale))
    p_w = wilson_log_prob(z, centric, multiplicity, sigma)
    return torch.where(root, p_w, p_dw)


# --------------------------------------------------------------------------------------------------
# scaling model  (careless/models/scaling/nn.py, image.py)
# --------------------------------------------------------------------------------------------------
def mlp_identity_init(d: int, width: int, n_layers: int, dtype=np.float32):
    """Identity-initialised Dense kernels (tf.eye(rows, cols), also when non-square [3P]) and zero biases:
    `MetadataScaler.__init__` (nn.py:48-79).  Returns (weights, biases) with the final Dense(2) last."""
    ws, bs = [], []
    fan_in = d
    for _ in range(n_layers):
        ws.append(np.eye(fan_in, width, dtype=dtype))
        bs.append(np.zeros(width, dtype=dtype))
        fan_in = width
    ws.append(np.eye(fan_in, 2, dtype=dtype))
    bs.append(np.zeros(2, dtype=dtype))
    return ws, bs


def _leaky_relu(z, leakiness: float, layer: int, flips, near, near_k: int):
    """LeakyReLU of the pre-activations `z` (N, w) of hidden layer `layer`.  Test hooks (the reference has neither):
    `flips` -- a collection of (layer, row, unit): those units take the OTHER branch than the sign of z says.  A pre-activation
    that lies within fp32 rounding of zero comes out on either side in an fp32 implementation; the value hardly moves
    (z ~ 0) but the derivative does (1 vs leakiness), so an fp32 engine is compared with the oracle under the engine's branches.
    `near` -- a list that receives (|z| / bound, layer, row, unit) of the pre-activations whose magnitude is below `bound`, the
    rounding error of their fp32 dot product (see `mlp_forward`)."""
    if near is not None:
        absz, bound = z.detach().abs(), near_k
        idx = torch.nonzero(absz <= bound)
        for r, u in idx.tolist():
            near.append((float(absz[r, u] / bound[r, u]), layer, r, u))
    if not flips or not any(f[0] == layer for f in flips):
        return torch.nn.functional.leaky_relu(z, negative_slope=leakiness)
    pos = z > 0
    flip = torch.zeros_like(pos)
    for l, r, u in flips:
        if l == layer:
            flip[r, u] = True
    return torch.where(pos ^ flip, z, leakiness * z)


def mlp_forward(metadata, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], leakiness: float,
                image_id=None, imgl_w: Sequence[torch.Tensor] = (), imgl_b: Sequence[torch.Tensor] = (), flips=None, near=None):
    """`MetadataScaler.call` (nn.py:92-103): L x Dense(w, LeakyReLU) then Dense(2, linear).  Returns (N, 2).
    With `imgl_w` / `imgl_b` it is `NeuralImageScaler.call` (image.py:116-125): after the Dense stack, one
    `ImageLayer` (image.py:90-96) per entry: act(matmul(w[image_id], h[..., None])[..., 0] + b[image_id]).
    `flips` / `near`: test hooks, see `_leaky_relu`; hidden layers are numbered through the Dense stack, then the image layers.
    The bound of `near` is 4 sqrt(k) eps32 (|h| |W| + |b|), k = the layer's fan-in: the statistical size of the rounding error of an
    fp32 dot product with a safety factor, not its worst case."""
    eps32 = float(np.finfo(np.float32).eps)
    h, layer = metadata, 0
    for w, b in zip(weights[:-1], biases[:-1]):
        z = h @ w + b
        bound = None
        if near is not None:
            bound = 4.0 * w.shape[0] ** 0.5 * eps32 * (h.detach().abs() @ w.detach().abs() + b.detach().abs())
        h = _leaky_relu(z, leakiness, layer, flips, near, bound)
        layer += 1
    for w, b in zip(imgl_w, imgl_b):
        z = torch.einsum("noi,ni->no", w[image_id], h) + b[image_id]
        bound = None
        if near is not None:
            bound = 4.0 * w.shape[-1] ** 0.5 * eps32 * (torch.einsum("noi,ni->no", w[image_id].detach().abs(), h.detach().abs()) + b[image_id].detach().abs())
        h = _leaky_relu(z, leakiness, layer, flips, near, bound)
        layer += 1
    return h @ weights[-1] + biases[-1]


def scale_bijector(raw, kind: str, epsilon: float):
    """`NormalLayer.call` (nn.py:22-25) with the CLI's bijector chains (manager.py:450-463):
    exp -> Shift(eps)(Exp(raw)); softplus -> Shift(eps)(Softplus(raw))."""
    if kind == "exp":
        return torch.exp(raw) + epsilon
    if kind == "softplus":
        return torch.nn.functional.softplus(raw) + epsilon
    raise ValueError(f"Unsupported scale bijector type, {kind}")


def image_scales(img_raw: torch.Tensor) -> torch.Tensor:
    """`ImageScaler.scales` (image.py:23-25): concat([1.], trainable (M-1,))."""
    one = torch.ones(1, dtype=img_raw.dtype)
    return torch.cat([one, img_raw])


# --------------------------------------------------------------------------------------------------
# likelihoods  (careless/models/likelihoods/mono.py, laue.py)
# --------------------------------------------------------------------------------------------------
def normal_log_prob(x, loc, scale):
    """`tfd.Normal.log_prob` [3P] (mono.py:16-18)."""
    y = (x - loc) / scale
    return -0.5 * y * y - 0.5 * LOG_2PI - torch.log(scale)


def studentt_log_prob(x, df: float, loc, scale):
    """`tfd.StudentT.log_prob` [3P] (mono.py:25-37):
    -0.5 (df+1) log1p(y^2/df) - log|scale| - 0.5 log df - 0.5 log pi - lgamma(df/2) + lgamma((df+1)/2)."""
    y = (x - loc) / scale
    return (-0.5 * (df + 1.0) * torch.log1p(y * y / df) - torch.log(torch.abs(scale)) - 0.5 * math.log(df)
            - 0.5 * math.log(math.pi) - math.lgamma(0.5 * df) + math.lgamma(0.5 * (df + 1.0)))


def laue_convolve(value, harmonic_id):
    """`ConvolvedLikelihood.convolve` (laue.py:17-25): scatter_nd(harmonic_id, value^T, shape (N,S))^T -- duplicates sum,
    untouched slots stay 0.  value: (S, N) or (N,)."""
    out = torch.zeros_like(value)
    return out.index_add(-1, harmonic_id, value)


# --------------------------------------------------------------------------------------------------
# the ELBO  (careless/models/merging/variational.py:141-183, 123-139)
# --------------------------------------------------------------------------------------------------
@dataclass
class ElboParams:
    q_loc_raw: torch.Tensor            # (R,)   log(loc)
    q_scale_raw: torch.Tensor          # (R,)   log(scale - eps)
    mlp_w: List[torch.Tensor]          # L x (in, w) + (w, 2)
    mlp_b: List[torch.Tensor]
    img_raw: Optional[torch.Tensor] = None   # (M-1,)
    dw_r_raw: Optional[torch.Tensor] = None  # (n_asu,) pre-sigmoid, only with --optimize-double-wilson-r
    ev11_raw: Optional[torch.Tensor] = None  # (3,) pre-softplus Sdfac, Sdadd, SdB (mono.py:42-44), only with cfg.ev11
    imgl_w: Optional[List[torch.Tensor]] = None   # K x (M, w, w) per-image kernels, only with cfg.image_layers (image.py:76-82)
    imgl_b: Optional[List[torch.Tensor]] = None   # K x (M, w)

    def tensors(self) -> List[torch.Tensor]:
        """Trainable tensors in the flat-buffer order the HIP engine uses."""
        out = [self.q_loc_raw, self.q_scale_raw]
        for w, b in zip(self.mlp_w, self.mlp_b):
            out += [w, b]
        if self.img_raw is not None:
            out.append(self.img_raw)
        for w, b in zip(self.imgl_w or [], self.imgl_b or []):
            out += [w, b]
        if self.ev11_raw is not None:
            out.append(self.ev11_raw)
        if self.dw_r_raw is not None:
            out.append(self.dw_r_raw)
        return out

    def clone(self, dtype=None, requires_grad=False) -> "ElboParams":
        def c(t):
            if t is None:
                return None
            t = t.detach().clone()
            if dtype is not None:
                t = t.to(dtype)
            return t.requires_grad_(requires_grad)
        return ElboParams(c(self.q_loc_raw), c(self.q_scale_raw), [c(w) for w in self.mlp_w],
                          [c(b) for b in self.mlp_b], c(self.img_raw), c(self.dw_r_raw), c(self.ev11_raw),
                          None if self.imgl_w is None else [c(w) for w in self.imgl_w],
                          None if self.imgl_b is None else [c(b) for b in self.imgl_b])


@dataclass
class ElboInputs:
    """The `inputs` tuple in `BaseModel.input_index` order (models/base.py:22-31), squeezed to 1-D, plus the
    per-reflection constants the prior and q need (io/manager.py:54-68, 432-436)."""
    refl_id: torch.Tensor              # (N,) int64
    image_id: torch.Tensor             # (N,) int64
    metadata: torch.Tensor             # (N, d)
    iobs: torch.Tensor                 # (N,)
    sigiobs: torch.Tensor              # (N,)
    centric: torch.Tensor              # (R,) bool
    multiplicity: torch.Tensor         # (R,) float
    low: torch.Tensor                  # (R,) float: 1e-32 * ~centric (manager.py:434)
    sigma: torch.Tensor = None         # () or (R,) Wilson Sigma (manager.py:43-52)
    harmonic_id: Optional[torch.Tensor] = None   # (N,) int64, Laue only
    # double-Wilson (wilson.py:82-138)
    parent_ids: Optional[torch.Tensor] = None
    root: Optional[torch.Tensor] = None
    asu_ids: Optional[torch.Tensor] = None
    dw_r: Optional[torch.Tensor] = None


def elbo_forward(p: ElboParams, x: ElboInputs, cfg: ElboConfig, u_f: torch.Tensor, eta: torch.Tensor,
                 kl_mask: Optional[torch.Tensor] = None, flips=None, near=None):
    """One forward pass of `VariationalMergingModel.call` (variational.py:141-183).

    u_f: (S, R) uniforms for the truncated normal; eta: (S, N) standard normals for the scale sample.
    kl_mask: optional (R,) bool -- restrict the KL sum to these reflections (used by the data-parallel shard
    tests so the KL is counted once across ranks).
    Returns dict(loss, nll, kl, ipred, z_f)."""
    S = cfg.mc_samples
    loc, scale = tn_loc_scale(p.q_loc_raw, p.q_scale_raw, cfg.epsilon)
    high = torch.as_tensor(cfg.high, dtype=loc.dtype)
    z_f = tn_sample(loc, scale, x.low, high, u_f)                              # variational.py:154

    out = mlp_forward(x.metadata, p.mlp_w, p.mlp_b, cfg.leakiness,            # variational.py:156 -> nn.py:106-120
                      x.image_id, p.imgl_w or (), p.imgl_b or (),             # --image-layers: image.py:116-125
                      flips=flips, near=near)                                  # (test hooks: LeakyReLU branches at fp32 rounding)
    s_loc = out[:, 0]
    s_sig = scale_bijector(out[:, 1], cfg.scale_bijector, cfg.epsilon)
    z_scale = s_loc[None, :] + s_sig[None, :] * eta + cfg.scale_shift          # variational.py:157; tfb.Shift(istd) nn.py:84-87
    if cfg.use_image_scales:
        a = image_scales(p.img_raw)[x.image_id]                                # image.py:40-42
        z_scale = a[None, :] * z_scale                                         # image.py:60-63 (Scale bijector)

    ipred = z_scale * z_f[:, x.refl_id] ** 2                                   # variational.py:167

    if cfg.laue:
        ipred_l = laue_convolve(ipred, x.harmonic_id)                          # laue.py:33-34
    else:
        ipred_l = ipred
    sig_l = x.sigiobs[None, :]
    if cfg.ev11:                                                                # Ev11Likelihood.corrected_sigiobs (mono.py:51-59)
        sd = torch.nn.functional.softplus(p.ev11_raw)
        sp = torch.nn.functional.softplus(ipred_l)
        sig_l = sd[0] * torch.sqrt(x.sigiobs[None, :] ** 2 + sd[2] * sp + sd[1] * sp * sp)
    if cfg.likelihood == "normal":
        ll = normal_log_prob(ipred_l, x.iobs[None, :], sig_l)
    elif cfg.likelihood == "studentt":
        ll = studentt_log_prob(ipred_l, float(cfg.dof), x.iobs[None, :], sig_l)
    else:
        raise ValueError(cfg.likelihood)

    log_q = tn_log_prob(z_f, loc, scale, x.low, high)                          # variational.py:128
    if cfg.prior == "wilson":
        log_p = wilson_log_prob(z_f, x.centric, x.multiplicity, x.sigma)
    else:
        dw_r = torch.sigmoid(p.dw_r_raw) if p.dw_r_raw is not None else x.dw_r        # wilson.py:105-110
        log_p = double_wilson_log_prob(z_f, x.centric, x.multiplicity, x.sigma, x.parent_ids, x.root,
                                       x.asu_ids, dw_r)
    kl_e = log_q - log_p
    if kl_mask is not None:
        kl_e = kl_e[:, kl_mask]

    if cfg.kl_weight is None:                                                   # variational.py:172-174
        kl = kl_e.sum() / S
        nll = -(ll.sum() / S)
        loss = nll + kl
    else:                                                                       # variational.py:175-177
        kl = kl_e.mean()
        nll = -ll.mean()
        loss = nll + cfg.kl_weight * kl
    return dict(loss=loss, nll=nll, kl=kl, ipred=ipred, z_f=z_f)


def elbo_value_and_grads(p: ElboParams, x: ElboInputs, cfg: ElboConfig, u_f, eta, kl_mask=None, flips=None, near=None):
    """Loss + reverse-mode gradients of every trainable tensor (variational.py:197-202).  `flips` / `near`: see `_leaky_relu`."""
    q = p.clone(requires_grad=True)
    out = elbo_forward(q, x, cfg, u_f, eta, kl_mask, flips=flips, near=near)
    ts = q.tensors()
    grads = torch.autograd.grad(out["loss"], ts, allow_unused=True)
    grads = [torch.zeros_like(t) if g is None else g for g, t in zip(grads, ts)]
    return {k: v.detach() for k, v in out.items()}, grads


# --------------------------------------------------------------------------------------------------
# the training step  (variational.py:185-224; io/manager.py:494-501)
# --------------------------------------------------------------------------------------------------
def global_norm(grads: Sequence[torch.Tensor]) -> torch.Tensor:
    """`tf.linalg.global_norm` (variational.py:205) -- computed BEFORE the non-finite sanitise."""
    return torch.sqrt(sum((g * g).sum() for g in grads))


def clip_grads(grads: List[torch.Tensor], cfg: ElboConfig) -> List[torch.Tensor]:
    """tf_keras optimizer gradient clipping (manager.py:498-500 hands all three to `tfk.optimizers.Adam`) [3P-recall]:
    `_BaseOptimizer._clip_gradients` applies the FIRST active mode and returns -- clipnorm (per-tensor `tf.clip_by_norm`), else
    global_clipnorm (`tf.clip_by_global_norm`), else clipvalue (`tf.clip_by_value`); the constructor refuses clipnorm together with
    global_clipnorm.  So `--clipnorm X --clipvalue Y` clips by norm only.  (Rounds 1-3 applied the modes cumulatively; the
    two-flag case of scripts/replay_golden_in_reference.py settles it on the first machine with TensorFlow.)"""
    if cfg.clipnorm is not None and cfg.global_clipnorm is not None:
        raise ValueError("At most one of `clipnorm` and `global_clipnorm` can be set")
    if cfg.clipnorm is not None and cfg.clipnorm > 0:
        out = []
        for g in grads:
            n = torch.sqrt((g * g).sum())
            out.append(torch.where(n > cfg.clipnorm, g * (cfg.clipnorm / n), g))
        return out
    if cfg.global_clipnorm is not None and cfg.global_clipnorm > 0:
        n = global_norm(grads)
        sc = cfg.global_clipnorm / torch.maximum(n, torch.as_tensor(cfg.global_clipnorm, dtype=n.dtype))
        return [g * sc for g in grads]
    if cfg.clipvalue is not None and cfg.clipvalue > 0:
        return [torch.clamp(g, -cfg.clipvalue, cfg.clipvalue) for g in grads]
    return grads


@dataclass
class AdamState:
    m: List[torch.Tensor]
    v: List[torch.Tensor]
    t: int = 0

    @staticmethod
    def zeros_like(ts: Sequence[torch.Tensor]) -> "AdamState":
        return AdamState([torch.zeros_like(t) for t in ts], [torch.zeros_like(t) for t in ts], 0)


def adam_apply(ts: List[torch.Tensor], grads: List[torch.Tensor], st: AdamState, cfg: ElboConfig):
    """tf_keras `Adam.update_step` [3P]: t = iterations + 1; alpha = lr sqrt(1-b2^t)/(1-b1^t);
    m += (g - m)(1-b1); v += (g^2 - v)(1-b2); var -= m alpha / (sqrt(v) + eps)."""
    st.t += 1
    t = st.t
    alpha = cfg.learning_rate * math.sqrt(1.0 - cfg.beta_2 ** t) / (1.0 - cfg.beta_1 ** t)
    for i, (p, g) in enumerate(zip(ts, grads)):
        st.m[i] = st.m[i] + (g - st.m[i]) * (1.0 - cfg.beta_1)
        st.v[i] = st.v[i] + (g * g - st.v[i]) * (1.0 - cfg.beta_2)
        p.data = p.data - st.m[i] * alpha / (torch.sqrt(st.v[i]) + cfg.adam_epsilon)


def train_step(p: ElboParams, x: ElboInputs, cfg: ElboConfig, st: AdamState, u_f, eta):
    """`train_step_with_gradient_norm` (variational.py:185-224): grads -> global norm -> non-finite -> 0 -> Adam.
    Mutates `p` and `st`; returns the metrics dict of that step."""
    out, grads = elbo_value_and_grads(p, x, cfg, u_f, eta)
    gnorm = global_norm(grads)
    grads = [torch.where(torch.isfinite(g), g, torch.zeros_like(g)) for g in grads]     # variational.py:208
    grads = clip_grads(grads, cfg)
    adam_apply(p.tensors(), grads, st, cfg)
    return {"loss": float(out["loss"]), "F KLDiv": float(out["kl"]), "NLL": float(out["nll"]),
            "Grad Norm": float(gnorm)}


# --------------------------------------------------------------------------------------------------
# the output step right after the path  (variational.py:47-121, io/manager.py:188-236)
# --------------------------------------------------------------------------------------------------
def _scale_moments(p: ElboParams, x: ElboInputs, cfg: ElboConfig):
    """mean / stddev of `scaling_model(inputs)`: Normal(loc, sigma) shifted by istd (tfb.Shift, nn.py:84-87) and, with image
    scales, scaled by a = w[image_id] (tfb.Scale, image.py:60-63) => mean a (loc + shift), stddev |a| sigma [3P]."""
    out = mlp_forward(x.metadata, p.mlp_w, p.mlp_b, cfg.leakiness, x.image_id, p.imgl_w or (), p.imgl_b or ())
    mean = out[:, 0] + cfg.scale_shift
    sd = scale_bijector(out[:, 1], cfg.scale_bijector, cfg.epsilon)
    if cfg.use_image_scales:
        a = image_scales(p.img_raw)[x.image_id]
        mean, sd = a * mean, a.abs() * sd
    return mean, sd


def scale_mean_stddev(p: ElboParams, x: ElboInputs, cfg: ElboConfig):
    """`VariationalMergingModel.scale_mean_stddev` (variational.py:47-78); Laue: mean and variance convolved (:70-76)."""
    with torch.no_grad():
        mean, sd = _scale_moments(p, x, cfg)
        if cfg.laue:
            mean = laue_convolve(mean, x.harmonic_id)
            sd = torch.sqrt(laue_convolve(sd * sd, x.harmonic_id))
    return mean, sd


def prediction_mean_stddev(p: ElboParams, x: ElboInputs, cfg: ElboConfig):
    """`VariationalMergingModel.prediction_mean_stddev` (variational.py:80-121): <I> = <Sigma><F^2>,
    var(I) = <F^4><Sigma^2> - <I>^2, <F^4> from `moment_4(method='scipy')` = scipy.stats.truncnorm.moment(4)
    (surrogate_posteriors.py:75-83; high = inf, :85), Laue: both convolved before the square root (:113-119)."""
    from scipy.stats import truncnorm
    with torch.no_grad():
        loc, scale = tn_loc_scale(p.q_loc_raw, p.q_scale_raw, cfg.epsilon)
        high = torch.as_tensor(cfg.high, dtype=loc.dtype)
        smean, ssd = _scale_moments(p, x, cfg)
        f2 = tn_mean(loc, scale, x.low, high) ** 2 + tn_variance(loc, scale, x.low, high)
        iexp = smean * f2[x.refl_id]
        a = ((x.low - loc) / scale).numpy()
        f4 = torch.as_tensor(truncnorm.moment(4, a, np.inf, loc.numpy(), scale.numpy()))
        s2 = smean ** 2 + ssd ** 2
        ivar = f4[x.refl_id] * s2 - iexp * iexp
        if cfg.laue:
            iexp, ivar = laue_convolve(iexp, x.harmonic_id), laue_convolve(ivar, x.harmonic_id)
    return iexp, torch.sqrt(ivar)


def merged_results(p: ElboParams, x: ElboInputs, cfg: ElboConfig, max_intensity_snr: float = 1e-5):
    """`DataManager.get_results` numerics (io/manager.py:188-197): F, SigF = TN mean / stddev, I = SigF^2 + F^2,
    SigI = sqrt(max((I snr)^2, <F^4> - I^2))."""
    from scipy.stats import truncnorm
    with torch.no_grad():
        loc, scale = tn_loc_scale(p.q_loc_raw, p.q_scale_raw, cfg.epsilon)
        high = torch.as_tensor(cfg.high, dtype=loc.dtype)
        F = tn_mean(loc, scale, x.low, high)
        SigF = torch.sqrt(tn_variance(loc, scale, x.low, high))
        I = SigF * SigF + F * F
        a = ((x.low - loc) / scale).numpy()
        f4 = torch.as_tensor(truncnorm.moment(4, a, np.inf, loc.numpy(), scale.numpy()))
        SigI = torch.sqrt(torch.maximum((I * max_intensity_snr) ** 2, f4 - I * I))
    return dict(F=F, SigF=SigF, I=I, SigI=SigI)


def validation_nll(p: ElboParams, x_val: ElboInputs, cfg: ElboConfig, u_f, eta, n_train: int):
    """What `train_model` logs as NLL_val (variational.py:248-260): the "NLL" metric of `test_on_batch(validation_data)`
    -- `call` on the validation tuple, so with `kl_weight` the mean runs over the VALIDATION observations (:175-177) --
    times len(train) / len(validation)."""
    with torch.no_grad():
        out = elbo_forward(p, x_val, cfg, u_f, eta)
    return float(out["nll"]) * n_train / int(x_val.refl_id.shape[0])


# --------------------------------------------------------------------------------------------------
# synthetic problems: the deterministic generator of SURVEY 8(d) lives in careless_amd/synthetic.py (plain numpy,
# no compute path) so that bench.py can build its workload without touching the oracle; re-exported here.
# --------------------------------------------------------------------------------------------------
from careless_amd.synthetic import (make_synthetic, make_synthetic_double_wilson, make_synthetic_laue,  # noqa: E402,F401
                                    positional_encoding, standardize_metadata)


def inputs_from_numpy(d: Dict, dtype=torch.float64, sigma=1.0) -> ElboInputs:
    centric = torch.as_tensor(np.asarray(d["centric"], dtype=bool))
    low = (1e-32 * (~np.asarray(d["centric"], dtype=bool))).astype(np.float32)     # manager.py:434
    kw = {}
    if "harmonic_id" in d and d["harmonic_id"] is not None:
        kw["harmonic_id"] = torch.as_tensor(np.asarray(d["harmonic_id"]).reshape(-1), dtype=torch.int64)
    for k in ("parent_ids", "asu_ids"):
        if k in d and d[k] is not None:
            kw[k] = torch.as_tensor(np.asarray(d[k]).reshape(-1), dtype=torch.int64)
    if "root" in d and d["root"] is not None:
        kw["root"] = torch.as_tensor(np.asarray(d["root"], dtype=bool))
    if "dw_r" in d and d["dw_r"] is not None:
        kw["dw_r"] = torch.as_tensor(np.asarray(d["dw_r"]), dtype=dtype)
    sig = torch.as_tensor(np.asarray(sigma, dtype=np.float32)).to(dtype)
    return ElboInputs(
        refl_id=torch.as_tensor(np.asarray(d["refl_id"]).reshape(-1), dtype=torch.int64),
        image_id=torch.as_tensor(np.asarray(d["image_id"]).reshape(-1), dtype=torch.int64),
        metadata=torch.as_tensor(np.asarray(d["metadata"], dtype=np.float32)).to(dtype),
        iobs=torch.as_tensor(np.asarray(d["iobs"], dtype=np.float32).reshape(-1)).to(dtype),
        sigiobs=torch.as_tensor(np.asarray(d["sigiobs"], dtype=np.float32).reshape(-1)).to(dtype),
        centric=centric,
        multiplicity=torch.as_tensor(np.asarray(d["multiplicity"], dtype=np.float32)).to(dtype),
        low=torch.as_tensor(low).to(dtype),
        sigma=sig,
        **kw,
    )


def init_params(d: Dict, cfg: ElboConfig, n_layers: int, width: Optional[int], dtype=torch.float64,
                init_scale: float = 1.0, sigma=1.0, rng: Optional[np.random.Generator] = None,
                perturb: float = 0.0) -> ElboParams:
    """Initial parameters as `DataManager.build_model` makes them (manager.py:432-436, 445-489):
    q = TruncatedNormal.from_loc_and_scale(prior.mean(), prior.stddev() * init_scale, low, scale_shift=eps),
    identity MLP, image scales 1.  `perturb` > 0 adds Gaussian noise to every parameter (in fp32, so the fp32 engine
    sees bit-identical values) so tests exercise non-trivial weights."""
    centric = np.asarray(d["centric"], dtype=bool)
    mult = np.asarray(d["multiplicity"], dtype=np.float32)
    # prior.mean()/stddev() are float32 tensors in the reference; mirror that rounding
    loc0 = wilson_mean(centric, mult, sigma).astype(np.float32)
    sc0 = (wilson_stddev(centric, mult, sigma).astype(np.float32) * np.float32(init_scale)).astype(np.float32)
    a, b = tn_raw_from_loc_scale(loc0, sc0, cfg.epsilon)
    dd = np.asarray(d["metadata"]).shape[-1]
    w = width if width is not None else dd
    ws, bs = mlp_identity_init(dd, w, n_layers)
    M = int(d["n_images"])
    img = np.ones(M - 1, dtype=np.float32) if cfg.use_image_scales else None
    arrs = [a.astype(np.float32), b.astype(np.float32)] + ws + bs + ([img] if img is not None else [])
    if perturb > 0.0:
        rng = rng or np.random.default_rng(0)
        a32 = (a + perturb * rng.normal(size=a.shape)).astype(np.float32)
        b32 = (b + perturb * rng.normal(size=b.shape)).astype(np.float32)
        ws = [(w_ + perturb * rng.normal(size=w_.shape)).astype(np.float32) for w_ in ws]
        bs = [(b_ + perturb * rng.normal(size=b_.shape)).astype(np.float32) for b_ in bs]
        if img is not None:
            img = (img + perturb * rng.normal(size=img.shape)).astype(np.float32)
    else:
        a32, b32 = a.astype(np.float32), b.astype(np.float32)
    t = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32)).to(dtype)
    ev = None
    if cfg.ev11:                       # TransformedVariable(1., Softplus()) x 3 (mono.py:42-44)
        ev = np.full(3, math.log(math.e - 1.0), dtype=np.float32)
        if perturb > 0.0:
            ev = (ev + perturb * rng.normal(size=3)).astype(np.float32)
    dwr = None
    if getattr(cfg, "optimize_dw_r", False):     # TransformedVariable(r, Sigmoid()): raw = logit(r); logit(0) = -inf for the root
        with np.errstate(divide="ignore"):
            r0 = np.asarray(d["dw_r"], dtype=np.float64)
            dwr = (np.log(r0) - np.log1p(-r0)).astype(np.float32)
    iw = ib = None
    if getattr(cfg, "image_layers", 0) > 0:       # ImageLayer: eye per image, zero bias (image.py:73-88)
        iw = [np.tile(np.eye(w, dtype=np.float32), (M, 1, 1)) for _ in range(cfg.image_layers)]
        ib = [np.zeros((M, w), dtype=np.float32) for _ in range(cfg.image_layers)]
        if perturb > 0.0:
            iw = [(v + perturb * rng.normal(size=v.shape)).astype(np.float32) for v in iw]
            ib = [(v + perturb * rng.normal(size=v.shape)).astype(np.float32) for v in ib]
    return ElboParams(t(a32), t(b32), [t(w_) for w_ in ws], [t(b_) for b_ in bs],
                      t(img) if img is not None else None, t(dwr) if dwr is not None else None,
                      t(ev) if ev is not None else None,
                      None if iw is None else [t(v) for v in iw], None if ib is None else [t(v) for v in ib])
