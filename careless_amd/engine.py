"""Host side of the MI355X ELBO engine: turns the plugin objects of a `VariationalMergingModel` plus the reference's
`inputs` tuple into device buffers, and enqueues one ELBO training step as a fixed sequence of C-ABI calls.

What one step replaces in the reference: `train_step_with_gradient_norm` (careless/models/merging/variational.py:185-224)
= forward (`call`, :141-183), `tape.gradient`, `tf.linalg.global_norm`, non-finite sanitise, Adam.

Step schedule (all on one HIP stream, no host synchronisation):
    cl_tn_forward (clears the step's accumulators on its way) -> cl_elbo_mono_fwd_bwd -> cl_tn_backward (carries cl_reduce_partials)
    -> [cl_owner_qnorm + all-reduce, data-parallel only: the flat gradient (row split) or its tail + 4 norm terms (reflection-owner
    split)] -> [cl_grad_sqnorm: clip modes only] -> cl_adam_step -> cl_step_finalize

HBM layout owned by the engine (see include/careless_hip.h):
    params / m / v / grads : one flat fp32 vector  [ q_loc_raw (R) | q_scale_raw (R) | scaler W^T layout (P) | image scales (M-1) | ... ]
    workspace              : [ dz_f (R*S) | grads (n) + message tail | scalars (4 doubles) | per-tensor norms | owner-mode scratch ]
    observation shard      : refl_id i32, image_id i32, meta_t [d][n_pad], iobs, sig [, noise_row]   (immutable)
PyTorch is used for device memory, streams and torch.distributed only.

Round 4: the observation layouts and the data-parallel splits live in careless_amd/obs.py, the layer-by-layer path of scalers wider
than 64 in careless_amd/wide.py (a mixin of `ElboEngine`); this file keeps the flat parameter layout, the step schedule and the helpers
behind the plugin protocol methods.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from careless_amd import _lib
from careless_amd._lib import AdamArgs, DetArgs, FrozenArgs, LaueArgs, MlpArgs, TnArgs, check, ptr
from careless_amd.models.base import BaseModel
# (re-exported: the tests, bench.py and scripts import the sharding / layout names from here)
from careless_amd.obs import (GRANULE, ObsChunks, ObsData, Shard, _EmptyObs, _ShardRows, _np, laue_group_shard, launch_row_limit,  # noqa: F401
                              make_shard, owner_bounds, owner_shard, pack_by_image, pack_laue)
from careless_amd.wide import WidePath

TILE = _lib.CL_MLP_TILE
HISTORY_KEYS = ("loss", "F KLDiv", "NLL", "Grad Norm")


def require_gpu(what: str) -> torch.device:
    """The product has no CPU path: every compute entry point calls this first."""
    _lib.get_lib()
    if not torch.cuda.is_available():
        raise _lib.CarelessHipError(f"{what} needs an AMD GPU (gfx950); careless_amd has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------------------------------------------------
# layout of the flat parameter vector
# ------------------------------------------------------------------------------------------------------------
@dataclass
class FlatLayout:
    R: int
    P: int
    n_img: int                 # M - 1 trainable image scales (0 when image scales are off)
    seg_off: List[int]         # tensor boundaries (Keras trainable-variable granularity) for per-tensor clipnorm
    seg_owner: List[str]       # "q" | "scaler" | "likelihood" per tensor (for --freeze-*)
    n_ev11: int = 0            # 3 with the Evans-2011 error model
    n_dwr: int = 0             # number of ASUs with --optimize-double-wilson-r
    n_imgl: int = 0            # floats of the per-image layers (NeuralImageScaler): K * M * (w*w + w)

    @property
    def n(self) -> int:
        return 2 * self.R + self.P + self.n_img + self.n_imgl + self.n_ev11 + self.n_dwr

    @property
    def off_dwr(self) -> int:
        return 2 * self.R + self.P + self.n_img + self.n_imgl + self.n_ev11

    @property
    def off_ev11(self) -> int:
        return 2 * self.R + self.P + self.n_img + self.n_imgl

    @property
    def off_imgl(self) -> int:
        return 2 * self.R + self.P + self.n_img

    @property
    def off_mlp(self) -> int:
        return 2 * self.R

    @property
    def off_img(self) -> int:
        return 2 * self.R + self.P


def make_layout(R: int, d: int, w: int, L: int, n_img: int, n_ev11: int = 0, n_dwr: int = 0, imgl=None) -> FlatLayout:
    """`imgl` = (K, M): K per-image layers over M images (NeuralImageScaler)."""
    seg, owner = [0, R, 2 * R], ["q", "q"]
    off, fan_in = 2 * R, d
    for _ in range(L):
        off += w * fan_in; seg.append(off); owner.append("scaler")
        off += w; seg.append(off); owner.append("scaler")
        fan_in = w
    off += 2 * fan_in; seg.append(off); owner.append("scaler")
    off += 2; seg.append(off); owner.append("scaler")
    P = off - 2 * R
    if n_img > 0:
        off += n_img; seg.append(off); owner.append("scaler")
    n_imgl = 0
    if imgl is not None:
        K, M = imgl
        for _ in range(K):               # ImageLayer kernel (M, w, w) and bias (M, w) (image.py:76-88)
            off += M * w * w; seg.append(off); owner.append("scaler")
            off += M * w; seg.append(off); owner.append("scaler")
        n_imgl = K * M * (w * w + w)
    for _ in range(n_ev11):              # Sdfac, Sdadd, SdB: three scalar variables (mono.py:42-44)
        off += 1; seg.append(off); owner.append("likelihood")
    if n_dwr > 0:                        # the double-Wilson r vector (wilson.py:105-110)
        off += n_dwr; seg.append(off); owner.append("prior")
    return FlatLayout(R, P, n_img, seg, owner, n_ev11, n_dwr, n_imgl)


@dataclass
class LayerBlock:
    """One launch of a chained scaler: Dense layers [l0, l1) of the stack; the last block also carries the Dense(2) head."""
    l0: int
    l1: int
    d_in: int          # width of the block's input: the metadata width for the first block, the hidden width for the others
    off: int           # offset of the block's parameters inside the scaler's flat W^T layout
    P: int             # number of parameters of the block (what its gradient partials hold)
    final: bool


def chain_plan(d: int, w: int, L: int, max_layers: int, last: int = 0) -> List[LayerBlock]:
    """Split a scaler of L Dense layers into the fewest, evenly sized blocks of at most `max_layers` layers (one block = one
    launch of the fused kernel, include/careless_hip.h: act_out / dH_ext / dX_out).  `last` > 0: the last block has exactly that many
    layers (round 6: 20 of them on the default scaler's lane kernel), the layers in front of it are split evenly."""
    if last > 0 and L > last:
        K = -(-(L - last) // max_layers)
        sizes = [(L - last) // K + (1 if i < (L - last) % K else 0) for i in range(K)] + [last]
        K += 1
    else:
        K = -(-L // max_layers)
        sizes = [L // K + (1 if i < L % K else 0) for i in range(K)]
    off_of = lambda l: 0 if l == 0 else w * d + w + (l - 1) * (w * w + w)
    blocks, l0 = [], 0
    for k, n in enumerate(sizes):
        l1 = l0 + n
        final = k == K - 1
        P = off_of(l1) - off_of(l0) + (2 * w + 2 if final else 0)
        blocks.append(LayerBlock(l0, l1, d if k == 0 else w, off_of(l0), P, final))
        l0 = l1
    return blocks



# ------------------------------------------------------------------------------------------------------------
# the engine
# ------------------------------------------------------------------------------------------------------------
class ElboEngine(WidePath):
    FROZEN_LAUE_PACKED = True        # ... harmonic groups: packed order + the two-call form of `cl_frozen_rows` (tests flip it to compare with the three slot launches)
    FROZEN_SORTED_ROWS = True        # a frozen scaler's monochromatic rows: sorted by reflection, `cl_frozen_rows` (round 6; tests flip it to compare with `cl_slot_rows`)
    SLOT_ROWS_ONE_LAUNCH = True      # rows that are their own slot: predict + log-prob + gradient in one `cl_slot_rows` launch (tests flip it)
    def __init__(self, model, inputs, seed: int = 1234, shard: Optional[Shard] = None, process_group=None,
                 grid: Optional[int] = None):
        self.device = require_gpu("ElboEngine")
        self.lib = _lib.get_lib()
        self.model = model
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.process_group = process_group
        # CARELESS_FORCE_DIST=1 (careless.py: _data_parallel): a one-rank run makes every collective call of the multi-GPU step
        if os.environ.get("CARELESS_FORCE_DIST", "0") == "1":
            import torch.distributed as dist
            self.force_allreduce = dist.is_available() and dist.is_initialized()
        dev = self.device

        from careless_amd.models.merging.surrogate_posteriors import TruncatedNormal
        from careless_amd.models.priors.wilson import DoubleWilsonPrior, WilsonPrior
        from careless_amd.models.likelihoods.mono import LocationScaleLikelihood
        from careless_amd.models.scaling.image import HybridImageScaler, NeuralImageScaler
        from careless_amd.models.scaling.nn import MetadataScaler

        q, prior, lik, scaler = model.surrogate_posterior, model.prior, model.likelihood, model.scaling_model
        if not isinstance(q, TruncatedNormal):
            raise NotImplementedError(f"surrogate posterior {type(q).__name__} is not supported by the HIP engine")
        if not isinstance(prior, (WilsonPrior, DoubleWilsonPrior)):
            raise NotImplementedError(f"prior {type(prior).__name__} is not supported by the HIP engine yet")
        from careless_amd.models.likelihoods.laue import LaueBase
        self.laue = BaseModel.is_laue(inputs)
        if self.laue != isinstance(lik, LaueBase):
            raise ValueError("Laue inputs (8-tuple with harmonic_id) need a careless_amd.models.likelihoods.laue likelihood, "
                             "monochromatic inputs a careless_amd.models.likelihoods.mono one")
        if not isinstance(lik, (LocationScaleLikelihood, LaueBase)) or lik.kind not in ("normal", "studentt"):
            raise NotImplementedError(f"likelihood {type(lik).__name__} is not supported by the HIP engine yet")
        imgl = None
        if isinstance(scaler, HybridImageScaler):
            mlp, img = scaler.mlp_scaler, scaler.image_scaler
        elif isinstance(scaler, NeuralImageScaler):
            mlp, img, imgl = scaler.metadata_scaler, None, scaler
        elif isinstance(scaler, MetadataScaler):
            mlp, img = scaler, None
        else:
            raise NotImplementedError(f"scaling model {type(scaler).__name__} is not supported by the HIP engine yet")
        if model.scale_prior is not None:
            raise NotImplementedError("scale_prior is never enabled by the reference CLI and is not supported")
        self.q, self.prior, self.lik, self.mlp, self.img, self.imgl = q, prior, lik, mlp, img, imgl

        # ---- observations -> device layout (built after the layout is known, below) ---------------------------
        self.R = int(q.loc_raw.numel())
        self.N_total = int(_np(BaseModel.get_refl_id(inputs)).reshape(-1).shape[0])
        self.d = int(_np(BaseModel.get_metadata(inputs)).reshape(self.N_total, -1).shape[1])
        self.shard = shard if shard is not None else make_shard(self.N_total, self.R)
        self.laue_groups = None
        if self.laue and self.shard.world > 1:
            self.laue_groups = laue_group_shard(_np(BaseModel.get_harmonic_id(inputs)).reshape(-1), self.shard.rank, self.shard.world)

        # ---- per-reflection constants ----------------------------------------------------------------
        self.low = q.low.to(dev, torch.float32).contiguous()
        self.centric = torch.as_tensor(prior.centric.astype(np.uint8), device=dev)
        self.es = torch.as_tensor(prior.eps_sigma, device=dev)
        if self.centric.numel() != self.R or self.es.numel() != self.R:
            raise ValueError("prior and surrogate posterior disagree on the number of reflections")
        self.double_wilson = isinstance(prior, DoubleWilsonPrior)
        if self.double_wilson:
            if prior.reflids.size != self.R or (prior.reflids >= self.R).any():
                raise ValueError("DoubleWilsonPrior.reflids does not match the surrogate posterior")
            self.parent_ids = torch.as_tensor(prior.reflids.astype(np.int32), device=dev)
            self.root = torch.as_tensor(prior.root.astype(np.uint8), device=dev)
            self.dw_r = torch.as_tensor(prior.r_per_reflection, device=dev)
            self.dw_children = None         # deterministic mode: the children of every reflection as a CSR list (built below)

        # ---- flat parameters; the plugin objects become views into them ----------------------------------
        mlp.build(self.d)
        self.w, self.L = mlp.width, mlp.n_layers
        n_img = 0
        if img is not None:
            n_img = img.max_images - 1
        self.ev11 = bool(getattr(lik, "ev11", False))
        self.dw_trainable = self.double_wilson and prior.r_raw is not None
        n_dwr = int(prior.r_raw.numel()) if self.dw_trainable else 0
        self.blocks = None
        self.chain_lane = False                      # (a chained scaler whose last 20 layers run on the lane kernel: set with the plan below)
        # hidden or metadata width beyond 64: the activations of a layer no longer fit a wave's registers next to the weight-gradient
        # blocks, so the scaler runs unfused -- one fp32-MFMA GEMM launch (csrc/wide_gemm.hip) per layer and direction, activations
        # through HBM -- around the same HIP likelihood kernels (_data_term_wide)
        self.wide = self.w > 64 or self.d > 64
        if imgl is not None and not self.wide and self.L + imgl.n_image_layers > self._imgl_layer_cap(imgl.n_image_layers):
            self.wide = True        # more hidden layers (Dense + per-image) than one fused launch holds: layer by layer as well
        max_plain = 1 if self.wide else int(self.lib.cl_mlp_max_layers(self.w))
        if not self.wide and imgl is None and self.L > max_plain:
            # deeper than one launch holds in registers: a chain of layer blocks, activations exchanged through HBM
            # width <= 10 (round 6): the LAST 20 layers and the head run on the lane-per-observation kernel -- the default scaler's own, with its
            # input = the activations of the block in front (w "metadata columns") and dZ_0 out; `cl_chain_dx` turns dZ_0 into the gradient
            # of those activations.  24 x 10 at 4 M observations: 3.13 -> 1.4 ms per step (two blocks of the 16-wide kernel before).
            # (widths 11, 12: the last NINETEEN layers -- the deepest twelve-wide instance without spilled registers -- and blocks of at most 19 in front)
            self.chain_lane = (self.w <= 12 and not self.laue and max_plain == 20 and os.environ.get("CARELESS_HIP_LANE", "1") != "0" and
                               os.environ.get("CARELESS_HIP_CHAIN_LANE", "1") != "0" and
                               (self.w <= 10 or os.environ.get("CARELESS_HIP_LANE_W12", "1") != "0"))
            last = (20 if self.w <= 10 else 19) if self.chain_lane else 0
            self.blocks = chain_plan(self.d, self.w, self.L, max_plain if last != 19 else 19, last=last)
        # The careless default scaler (20 layers, hidden width <= 10) on more metadata columns than its lane-per-observation kernel holds
        # (31; four positionally encoded keys give 37): the first Dense layer is "peeled" -- its pre-activations come from
        # cl_peel_forward, the fused kernel runs the same scaler with an identity first layer on them (w "metadata columns": the shape it
        # is fastest at) and hands back dZ_0, cl_peel_backward takes W_0's and b_0's gradient (csrc/elbo_peel.hip).  Training
        # launches only: forward-only / external-gradient launches (predictions, two-pass Laue) keep the original scaler.
        # The same for every other scaler of hidden width <= 15 on more than 15 columns: elbo_narrow.hip holds <= 15 of them.
        lane_shape = self.L == 20 and self.w <= 10 and os.environ.get("CARELESS_HIP_LANE", "1") != "0"
        # (widths 13 .. 15 -- the narrow kernel's four-step instance, ~90 spilled registers -- are no faster there than on the 16-wide
        #  instance of elbo_mlp.hip, which takes up to 64 columns itself: profiles/r5_envelope.txt, 20 x 15 on 37 columns 2.02 against 2.19 ms)
        self.peel = (not self.wide and self.blocks is None and imgl is None and self.w <= 12 and self.L <= 20 and
                     self.d > (31 if lane_shape else 15) and self.d > self.w and bool(self.lib.cl_peel_supported(self.d, self.w, self.L)) and
                     os.environ.get("CARELESS_HIP_NARROW", "1") != "0")
        # ... and (round 5) the default scaler with one or two per-image layers on more than the 15 columns its lane instances hold
        # (`--image-layers 2 --positional-encoding-keys X,Y`: 21): the peeled layer's w pre-activations are the lane kernel's "metadata"
        # (round 6: Laue data too -- the per-image-layer instances are packed-layout kernels either way, and their dZ_0-storing form is back)
        # (... and at 2 .. 19 Dense layers of width 5 .. 10: the per-depth units carry the per-image-layer instances as well)
        lane_imgl_shape = lane_shape or (2 <= self.L < 20 and 5 <= self.w <= 10 and os.environ.get("CARELESS_HIP_LANE", "1") != "0" and
                                         os.environ.get("CARELESS_HIP_LANE_DEPTHS", "1") != "0")
        lane_imgl_max = 2 if self.L == 19 else 3          # (csrc/elbo_lane.hip: CL_LANE_IMGL_MAX_NL, CL_LANE_IMGL3_DEPTH_MAX)
        if (not self.wide and imgl is not None and lane_imgl_shape and imgl.n_image_layers <= lane_imgl_max and self.d > 15 and self.d > self.w and
                bool(self.lib.cl_peel_supported(self.d, self.w, self.L))):
            self.peel = True
        if imgl is not None:
            imgl.build(self.d)
            max_l = self._imgl_layer_cap(imgl.n_image_layers)
            if not self.wide and self.L + imgl.n_image_layers > max_l:
                raise NotImplementedError(f"{self.L} Dense + {imgl.n_image_layers} image layers of width {self.w}: the HIP engine "
                                          f"supports {max_l} hidden layers in total at this width")
        # Deterministic mode (`model.deterministic = True` or CARELESS_HIP_DETERMINISTIC=1): no float atomics anywhere in the step --
        # per-observation stores + fixed-order sums (cl_det_reduce) -- so two runs give bit-identical gradients and parameters
        self.deterministic = bool(getattr(model, "deterministic", False)) or os.environ.get("CARELESS_HIP_DETERMINISTIC", "0") == "1"
        two_pass = self.laue and bool(getattr(model, "laue_two_pass", False))
        # (wide scalers: monochromatic rows only -- they are their own slots and one kernel holds every float atomic of the path --, with a
        #  sample count that divides 64, so that a row's samples sit inside one wave)
        wide_det_ok = self.wide and not self.laue and 64 % int(model.mc_sample_size) == 0
        # (per-image layers, round 6: where the step runs the lane kernel's per-image-layer instances -- one or two of them on 2 .. 20 Dense
        #  layers of width <= 10, up to 15 columns or behind the peeled first layer -- one wave holds all tiles of an image in this mode)
        imgl_det_ok = (imgl is not None and not self.wide and imgl.n_image_layers <= lane_imgl_max and lane_imgl_shape and (self.d <= 15 or self.peel) and
                       os.environ.get("CARELESS_HIP_LANE", "1") != "0")
        if self.deterministic and (two_pass or (self.wide and not wide_det_ok) or (imgl is not None and not imgl_det_ok) or
                                   (self.blocks is not None and self.laue) or (self.double_wilson and prior.r_raw is not None)):
            raise NotImplementedError("deterministic mode covers monochromatic and single-pass Laue data, the Wilson and the double-Wilson prior "
                                      "(fixed r), Normal / Student-T likelihoods with or without the Evans-2011 error model, scalers of any depth up to "
                                      "width 64, up to three per-image layers on 2 .. 20 Dense layers of width 5 .. 10 (two on 19; the default scaler's kernels) and, "
                                      "for monochromatic data with a sample count that divides 64, scalers wider than 64; the two-pass "
                                      "Laue path (also under a chained scaler), a trainable double-Wilson r and every other shape with per-image layers "
                                      "keep their float atomics")
        if self.deterministic and self.double_wilson:
            # parents pull their children's terms in list order instead of children scattering with atomics (cl_dw_prior_forward)
            par = np.asarray(prior.reflids).astype(np.int64)
            kids = np.nonzero(par >= 0)[0]
            kids = kids[np.argsort(par[kids], kind="stable")]
            seg = np.concatenate([[0], np.cumsum(np.bincount(par[kids], minlength=self.R))]).astype(np.int32)
            self.dw_children = (torch.as_tensor(seg, device=dev), torch.as_tensor(kids.astype(np.int32) if len(kids) else np.zeros(1, np.int32), device=dev))
        self.layout = make_layout(self.R, self.d, self.w, self.L, n_img, 3 if self.ev11 else 0, n_dwr,
                                  imgl=(imgl.n_image_layers, imgl.max_images) if imgl is not None else None)
        lay = self.layout
        assert lay.P == mlp.param_count(self.d) == int(self.lib.cl_mlp_param_count(self.d, self.w, self.L))
        self.params = torch.empty(lay.n, dtype=torch.float32, device=dev)
        self.params[0:self.R] = q.loc_raw.to(dev)
        self.params[self.R:2 * self.R] = q.scale_raw.to(dev)
        self.params[lay.off_mlp:lay.off_mlp + lay.P] = mlp.flat.to(dev)
        q.loc_raw = self.params[0:self.R]
        q.scale_raw = self.params[self.R:2 * self.R]
        q.low = self.low
        mlp.flat = self.params[lay.off_mlp:lay.off_mlp + lay.P]
        if n_img > 0:
            self.params[lay.off_img:lay.off_img + n_img] = img._scales.to(dev)
            img._scales = self.params[lay.off_img:lay.off_img + n_img]
        if imgl is not None:
            self.params[lay.off_imgl:lay.off_imgl + lay.n_imgl] = imgl.flat.to(dev)
            imgl.flat = self.params[lay.off_imgl:lay.off_imgl + lay.n_imgl]
        if self.ev11:
            self.params[lay.off_ev11:lay.off_ev11 + 3] = lik.raw.to(dev)
            lik.raw = self.params[lay.off_ev11:lay.off_ev11 + 3]
        if self.dw_trainable:
            self.params[lay.off_dwr:lay.off_dwr + n_dwr] = prior.r_raw.to(dev)
            prior.r_raw = self.params[lay.off_dwr:lay.off_dwr + n_dwr]
            self.asu_ids = torch.as_tensor(prior.asu_ids.astype(np.int32), device=dev)
        self.adam_m = torch.zeros_like(self.params)
        self.adam_v = torch.zeros_like(self.params)
        self.t = 0                                  # optimizer iterations
        self.seg_off = torch.as_tensor(np.asarray(lay.seg_off, dtype=np.int32), device=dev)
        self.nseg = len(lay.seg_off) - 1

        # ---- workspace -----------------------------------------------------------------------------------
        self.S = int(model.mc_sample_size)
        if self.S < 1:
            raise ValueError("mc_sample_size must be >= 1")
        # Reflection-owner sharding: with more than one rank, monochromatic data and the Wilson prior, a rank takes a
        # RANGE OF REFLECTIONS and every observation of theirs instead of a range of rows.  Sampling q(F), its KL, its gradient and
        # its Adam update are then local to the owner (1 / world of the replicated work of the row split), and the step's all-reduce
        # carries the scaler's gradient and four norm terms instead of 2 R floats.  Laue data (a harmonic group mixes reflections),
        # the double-Wilson prior (a child's parent may live on another rank), per-image layers, wide scalers and the deterministic
        # mode keep the row split.  OPT-IN (`model.owner_shard = True` or CARELESS_HIP_OWNER_SHARD=1) since round 5: the split has been
        # verified on one device and over gloo (parity suite, shard-sum tests) but has never met RCCL on a multi-GPU node, and on one
        # device its compute side is worth <= 1 % at eight ranks; the row split stays the default until
        # `scripts/scale_curve.sh` has measured both on a node.
        self.owner = False
        want = getattr(model, "owner_shard", None)
        if want is None:
            env = os.environ.get("CARELESS_HIP_OWNER_SHARD", "")
            want = env == "1"
        if (want and self.shard.world > 1 and not self.laue and not self.double_wilson and not self.wide and imgl is None
                and not self.deterministic and not self.shard.owner):
            osh = owner_shard(_np(BaseModel.get_refl_id(inputs)).reshape(-1), self.R, self.shard.rank, self.shard.world)
            if osh is not None:
                self.shard = osh
        self.owner = bool(self.shard.owner)
        # A frozen scaling model (`--freeze-scales`; the half-dataset trainings of `--merge-half-datasets`, reference careless.py:102-128):
        # its output (loc, sigma) per observation does not change from step to step, so a step needs neither its forward nor its backward
        # pass -- only the sampling / likelihood part (`_data_term_frozen`).  An engine built around a frozen scaler lays its observations out
        # for that (plain rows in the caller's order: no packing by image or harmonic group).
        want_ff = getattr(model, "frozen_scaler_fast_path", None)
        self.frozen_fast = (os.environ.get("CARELESS_HIP_FROZEN_FAST", "1") != "0") if want_ff is None else bool(want_ff)
        self._grid_arg = grid
        self._frozen_layout = self.frozen_fast and not model.scaling_model.trainable and not (self.deterministic and self.laue)
        self._frozen_epoch = 0
        self.scaler_frozen = False
        self.obs = self._build_obs(inputs, self.shard.start, self.shard.stop, grid, self.laue_groups, rows=self.shard.rows if self.owner else None)
        if self.blocks is not None:
            self.obs.alloc_chain(self.lib, self.blocks, self.w, dev)
        RS = self.R * self.S
        o_dz = 0
        # the buffer a step all-reduces starts on a 16-byte boundary: the flat gradient (row split), or -- reflection-owner split -- what
        # lies behind a and b in it (two floats of padding in front of the gradient when 2 R is not a multiple of four)
        pad = (-2 * self.R) % 4 if self.owner else 0
        o_g = (RS + 3) // 4 * 4 + pad
        o_sc = (o_g + lay.n + 8 + 3) // 4 * 4           # 4 floats of slack, then the 4 norm terms of the owner-mode message; 16-byte aligned
        o_seg = o_sc + 8                                # 4 doubles = 8 floats
        o_own = o_seg + 2 * self.nseg                   # owner mode: 4 double accumulators + the block ticket of cl_owner_qnorm
        tot = o_own + 10
        self.ws = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.dz_f = self.ws[o_dz:o_dz + RS]
        self.grads = self.ws[o_g:o_g + lay.n]
        self.scalars = self.ws[o_sc:o_sc + 8].view(torch.float64)
        self.seg_sq = self.ws[o_seg:o_seg + 2 * self.nseg].view(torch.float64)
        self.own_scratch = self.ws[o_own:o_own + 10].view(torch.float64)
        self.msg_norm = self.ws[o_g + lay.n + 4:o_g + lay.n + 8]                # [raw, sanitised, sanitised a, sanitised b]
        self.msg = self.ws[o_g + 2 * self.R:o_g + lay.n + 8]                    # what an owner-mode step all-reduces
        self.ws_step = self.ws[o_g - pad:]                                      # owner mode zeroes this (16-byte aligned) and its own slice of dz_f per step
        assert (self.msg if self.owner else self.grads).data_ptr() % 16 == 0 and self.ws_step.data_ptr() % 16 == 0
        self.norm_part = torch.zeros(2 * 1024, dtype=torch.float64, device=dev)                  # per-workgroup norm sums of cl_adam_step (<= 1024 workgroups)
        self.kl_part = torch.zeros((self.R + 255) // 256, dtype=torch.float64, device=dev)      # per-workgroup KL sums of cl_tn_forward
        self.kl_part_dw = torch.zeros_like(self.kl_part) if self.double_wilson else None        # ... and of cl_dw_prior_forward
        self.z_f = torch.empty(RS, dtype=torch.float32, device=dev)
        self.stop_flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.frozen = torch.zeros(self.nseg, dtype=torch.uint8, device=dev)
        self.history_buf: Optional[torch.Tensor] = None
        self._keep = None
        self.refresh_config()

    def _build_obs(self, inputs, start, stop, grid, laue_groups, rows=None):
        """Device image of rows [start, stop) of `inputs`: one `ObsData`, or -- plain layout only -- as many pieces as the 4-GiB
        bound of a launch asks for (`ObsChunks`)."""
        kw = dict(grid=grid, n_refl=self.R, n_images=self._max_images(), laue_groups=laue_groups, pack_images=self.imgl is not None and not self.wide,
                  sort_images=self.imgl is not None and self.wide,
                  laue_single_pass=not getattr(self.model, "laue_two_pass", False) and self.blocks is None and not self.wide, wide=self.wide)
        if getattr(self, "_frozen_layout", False):
            # plain rows for a frozen scaler (its output is a constant: no tile / image constraint) -- except Laue data, which keeps the packed
            # order of the single-pass kernels (a group inside a 16-row granule) for `cl_frozen_rows`' group sums (round 6)
            kw.update(pack_images=False, sort_images=False,
                      laue_single_pass=bool(self.laue and self.FROZEN_LAUE_PACKED and not self.deterministic and not getattr(self.model, "laue_two_pass", False)))
        n_total = int(_np(BaseModel.get_refl_id(inputs)).reshape(-1).shape[0])
        stop = n_total if stop is None else stop
        per = launch_row_limit(self.d, self.S if self.deterministic else 0)
        if rows is not None:
            start, stop = 0, len(rows)
        if self.laue or self.imgl is not None or self.wide or stop - start <= per:
            o = ObsData(self.lib, inputs, start, stop, self.S, self.layout.P, self.device, rows=rows, **kw)
            o.host_inputs = inputs                        # (a reference: the frozen-scaler path reads the rows' metadata once per training)
            if self.deterministic:
                self._det_attach(o, [o])
            return o
        pieces = []
        for a in range(start, stop, per):
            b = min(stop, a + per)
            pieces.append(ObsData(self.lib, inputs, a, b, self.S, self.layout.P, self.device, rows=None if rows is None else rows[a:b], **kw))
            pieces[-1].row0 = a - start                   # first row of the piece inside the shard's eta / ipred arrays
            pieces[-1].is_piece = True                    # (several launches share dz_f: `cl_frozen_rows` adds instead of storing)
            pieces[-1].host_inputs = inputs
            kw["n_refl"] = kw["n_images"] = None          # (the id ranges were checked over the whole input by the first piece)
            if len(pieces) > 1:
                pieces[-1].partials = pieces[0].partials  # launches are serialised on one stream: one partial buffer
        o = ObsChunks(pieces)
        if rows is not None:
            o.rows = np.asarray(rows, dtype=np.int64)
        if self.deterministic:
            self._det_attach(o, pieces)
        return o

    def _det_attach(self, obs, pieces):
        """Buffers of the deterministic mode for one observation set: the per-observation stores of the fused kernel and the sort
        orders (by reflection, by image; stable, so sums run in row order) that `cl_det_reduce` walks."""
        dev = self.device
        if pieces[0].row_map is not None:
            # packed layout (single-pass Laue; per-image layers, round 6): the kernels store by the caller's local row (row_map), so the ids go back to that order
            p0 = pieces[0]
            if self.laue and not p0.fused_laue:
                raise NotImplementedError("deterministic mode: harmonic groups of more than 16 rows take the two-pass Laue path, which keeps its float atomics")
            rm = p0.row_map.cpu().numpy()
            ok = rm >= 0
            rid, img = np.zeros(p0.N, dtype=np.int64), np.zeros(p0.N, dtype=np.int64)
            rid[rm[ok]] = p0.refl_id.cpu().numpy()[ok]
            img[rm[ok]] = p0.image_id.cpu().numpy()[ok]
        else:
            rid = torch.cat([p.refl_id[: p.N] for p in pieces]).cpu().numpy()
            img = torch.cat([p.image_id[: p.N] for p in pieces]).cpu().numpy()
        n = len(rid)

        def order(ids, nbins):
            perm = np.argsort(ids, kind="stable").astype(np.int32)
            seg = np.concatenate([[0], np.cumsum(np.bincount(ids, minlength=nbins))]).astype(np.int32)
            return torch.as_tensor(perm, device=dev), torch.as_tensor(seg, device=dev)
        M = self._max_images() or 1
        # slot[i] = position of observation i in the reflection-sorted order: the kernels store its record THERE (cl_mlp_args.det_slot),
        # so that the per-reflection sums read contiguous records instead of gathering ~30 random ones each
        perm_r, seg_r = order(rid, self.R)
        slot = None
        if 4 * n * self.S < (1 << 32):          # (the kernels address a record with a 32-bit byte offset from the buffer's start)
            slot = torch.empty(n, dtype=torch.int32, device=dev)
            slot[perm_r.long()] = torch.arange(n, dtype=torch.int32, device=dev)
        obs.det = dict(slot=slot, dzf=torch.zeros(n * self.S, dtype=torch.float32, device=dev), dimg=torch.zeros(n, dtype=torch.float32, device=dev),
                       nll=torch.zeros(len(pieces) * pieces[0].grid + (_lib.CL_LAUE_LIK_MAX_BLOCKS if (self.laue or self.wide) else 0), dtype=torch.float64,
                                       device=dev),
                       refl=(perm_r, seg_r), img=order(img, M), M=M, pieces=len(pieces), grid=pieces[0].grid)
        if self.ev11:
            # Evans-2011 gradients: one slot of three floats per wave of every launch (the fused launches' CL_EV11_WAVES per workgroup, then
            # four per workgroup of the slot / padded-slot likelihood launch), summed in slot order by cl_det_reduce
            obs.det["ev11"] = torch.zeros(3 * (_lib.CL_EV11_WAVES * len(pieces) * pieces[0].grid + 4 * _lib.CL_LAUE_LIK_MAX_BLOCKS), dtype=torch.float32, device=dev)
        for k, p in enumerate(pieces):
            p.det_parent, p.det_index = obs, k

    def _imgl_layer_cap(self, K: int) -> int:
        """Hidden layers (Dense + per-image) one fused training launch holds at this width.  Width <= 15 on more than 32 metadata columns
        takes the 32-wide instance (`cl_launch_mlp`: the 16-wide <16, 64, 24, image layers> instance is withdrawn -- csrc/elbo_mlp.hip,
        launch_mode) unless the step runs the lane kernel's per-image-layer instances behind a peeled first layer."""
        lane = os.environ.get("CARELESS_HIP_LANE", "1") != "0"
        depths = lane and os.environ.get("CARELESS_HIP_LANE_DEPTHS", "1") != "0"
        lane_route = ((self.L == 20 and self.w <= 10 and K <= 3 and lane) or
                      (2 <= self.L <= 19 and 5 <= self.w <= 10 and K <= (2 if self.L == 19 else 3) and depths))
        if self.w <= 15 and self.d > 32 and not lane_route:
            return int(self.lib.cl_mlp_max_layers_imgl(32))
        return int(self.lib.cl_mlp_max_layers_imgl(self.w))

    def _max_images(self):
        if self.img is not None:
            return self.img.max_images
        return self.imgl.max_images if self.imgl is not None else None

    # the training shard's arrays under their old names
    N = property(lambda self: self.obs.N)
    n_pad = property(lambda self: self.obs.n_pad)
    grid = property(lambda self: self.obs.grid)
    refl_id = property(lambda self: self.obs.refl_id)
    image_id = property(lambda self: self.obs.image_id)
    meta_t = property(lambda self: self.obs.meta_t)
    iobs = property(lambda self: self.obs.iobs)
    sig = property(lambda self: self.obs.sig)
    partials = property(lambda self: self.obs.partials)

    # ------------------------------------------------------------------------------------------------------
    def refresh_config(self):
        """(Re)read the knobs that may change between train_model calls: trainable flags, kl weight, optimizer."""
        m = self.model
        fr = np.zeros(self.nseg, dtype=np.uint8)
        for k, owner in enumerate(self.layout.seg_owner):
            if owner == "q" and not self.q.trainable:
                fr[k] = 1
            if owner == "scaler" and not m.scaling_model.trainable:
                fr[k] = 1
        self.any_frozen = bool(fr.any())
        self.frozen.copy_(torch.as_tensor(fr))
        self.scaler_frozen = not m.scaling_model.trainable
        self._frozen_epoch += 1                          # (loc, sigma) of a frozen scaler are taken again at every train_model call
        if self._frozen_layout and not self.scaler_frozen:
            # the scaler was frozen when the engine laid its observations out and is trainable now: the fused kernels' layouts again
            self._frozen_layout = False
            first = self.obs.children[0] if isinstance(self.obs, ObsChunks) else self.obs
            self.obs = self._build_obs(first.host_inputs, self.shard.start, self.shard.stop, self._grid_arg, self.laue_groups,
                                       rows=self.shard.rows if self.owner else None)
            if self.blocks is not None:
                self.obs.alloc_chain(self.lib, self.blocks, self.w, self.device)
        opt = m.optimizer
        self.opt = opt
        S, N, R = self.S, self.N_total, self.R
        if m.kl_weight is None:                         # variational.py:172-174
            self.w_ll, self.w_kl, self.kl_mult = 1.0 / S, 1.0 / S, 1.0
        else:                                           # variational.py:175-177
            self.w_ll, self.w_kl, self.kl_mult = 1.0 / (S * N), 1.0 / (S * R), float(m.kl_weight)
        if self.lik.kind == "studentt":
            nu = float(self.lik.dof)
            self.lik_kind, self.dof = _lib.CL_LIK_STUDENTT, nu
            self.lik_const = math.lgamma(0.5 * (nu + 1.0)) - math.lgamma(0.5 * nu) - 0.5 * math.log(nu * math.pi)
        else:
            self.lik_kind, self.dof, self.lik_const = _lib.CL_LIK_NORMAL, 0.0, 0.0
        self.bij_kind = _lib.CL_BIJ_EXP if self.mlp.scale_bijector == "exp" else _lib.CL_BIJ_SOFTPLUS

    # ------------------------------------------------------------------------------------------------------
    def _tn_args(self, step: int, u_f) -> TnArgs:
        lay = self.layout
        a = TnArgs()
        a.q_loc_raw = self.params.data_ptr()
        a.q_scale_raw = self.params.data_ptr() + 4 * self.R
        a.low = ptr(self.low); a.centric = ptr(self.centric); a.es = ptr(self.es)
        a.R, a.S = self.R, self.S
        a.high, a.eps = self.q.high, self.q.scale_shift
        a.w_kl, a.kl_grad_mult = self.w_kl, self.kl_mult
        a.kl_begin, a.kl_end = self.shard.kl_begin, self.shard.kl_end
        if self.owner:
            a.r_begin, a.r_end = self.shard.kl_begin, self.shard.kl_end      # the reflections this rank owns: nobody else touches them
        a.u_f = ptr(u_f)
        a.seed, a.step = self.seed, step & 0xFFFFFFFF
        a.z_f = ptr(self.z_f); a.dz_f = ptr(self.dz_f)
        a.d_loc_raw = self.grads.data_ptr()
        a.d_scale_raw = self.grads.data_ptr() + 4 * self.R
        a.scalars = ptr(self.scalars)
        a.stop_flag = ptr(self.stop_flag)
        if self.double_wilson:
            a.prior_kind = _lib.CL_PRIOR_DOUBLE_WILSON
            a.parent_ids, a.root, a.dw_r = ptr(self.parent_ids), ptr(self.root), ptr(self.dw_r)
            a.dz_f_out = ptr(self.dz_f)
            if self.dw_children is not None:
                a.dw_child_seg, a.dw_child_ids = ptr(self.dw_children[0]), ptr(self.dw_children[1])
            if self.dw_trainable:
                a.dw_r_raw = self.params.data_ptr() + 4 * lay.off_dwr
                a.d_dw_r_raw = self.grads.data_ptr() + 4 * lay.off_dwr
                a.asu_ids, a.n_asu = ptr(self.asu_ids), lay.n_dwr
        return a

    def _mlp_args(self, step: int, eta, ipred_out=None, obs: Optional[ObsData] = None) -> MlpArgs:
        lay = self.layout
        obs = self.obs if obs is None else obs
        a = MlpArgs()
        a.refl_id = ptr(obs.refl_id); a.image_id = ptr(obs.image_id); a.meta_t = ptr(obs.meta_t)
        a.iobs = ptr(obs.iobs); a.sig = ptr(obs.sig)
        a.n_obs, a.n_pad = obs.N, obs.n_pad
        a.obs_offset = obs.start
        if obs.row_map is not None:
            a.n_obs = obs.n_pad                         # packed: validity is per row (row_map / refl_id = -1)
            a.row_map = ptr(obs.row_map)
        if obs.fused_laue:
            a.gmeta, a.tile_gmax = ptr(obs.gmeta), ptr(obs.tile_gmax)
        if getattr(obs, "noise_row", None) is not None:
            a.noise_row = ptr(obs.noise_row)
        if self.imgl is not None:
            a.imgl = self.params.data_ptr() + 4 * lay.off_imgl
            a.d_imgl = self.grads.data_ptr() + 4 * lay.off_imgl
            a.n_imgl, a.n_images = self.imgl.n_image_layers, self.imgl.max_images
            a.tile_img = ptr(obs.tile_img)
        a.mlp = self.params.data_ptr() + 4 * lay.off_mlp
        a.d, a.w, a.L = self.d, self.w, self.L
        a.leak = self.mlp.leakiness
        a.use_img = 1 if lay.n_img > 0 else 0
        a.img = (self.params.data_ptr() + 4 * lay.off_img) if lay.n_img > 0 else None
        a.z_f = ptr(self.z_f)
        a.R, a.S = self.R, self.S
        a.lik_kind, a.dof, a.lik_const = self.lik_kind, self.dof, self.lik_const
        a.bij_kind, a.eps = self.bij_kind, self.mlp.epsilon
        a.shift = self.mlp.scale_multiplier or 0.0
        a.w_ll = self._w_ll(obs)
        row0 = getattr(obs, "row0", 0)                  # piece of a chunked shard: its rows inside the shard's eta / ipred arrays
        a.eta = ptr(eta) + 4 * self.S * row0 if eta is not None else None
        a.seed, a.step = self.seed, step & 0xFFFFFFFF
        a.dz_f = ptr(self.dz_f)
        a.d_img = (self.grads.data_ptr() + 4 * lay.off_img) if lay.n_img > 0 else None
        a.partials = ptr(obs.partials)
        a.scalars = ptr(self.scalars)
        a.ipred_out = ptr(ipred_out) + 4 * self.S * row0 if ipred_out is not None else None
        a.stop_flag = ptr(self.stop_flag)
        if self.ev11:
            a.ev11 = self.params.data_ptr() + 4 * lay.off_ev11
            a.d_ev11 = self.grads.data_ptr() + 4 * lay.off_ev11
        if self.deterministic:
            det = obs.det_parent.det
            if det["slot"] is not None:
                a.dzf_obs = det["dzf"].data_ptr()                  # (records by slot: positions inside the whole shard, not the piece)
                a.det_slot = det["slot"].data_ptr() + 4 * row0
            else:
                a.dzf_obs = det["dzf"].data_ptr() + 4 * self.S * row0
            a.dimg_obs = det["dimg"].data_ptr() + 4 * row0
            a.nll_part = det["nll"].data_ptr() + 8 * det["grid"] * obs.det_index
            if self.ev11:
                a.ev11_part = det["ev11"].data_ptr() + 4 * 3 * _lib.CL_EV11_WAVES * det["grid"] * obs.det_index
        return a

    def kernel_name(self, mode: int = 0) -> str:
        """Name of the kernel instance the scaler launches of this engine run (`cl_mlp_kernel_name`: the library's own routing):
        what a rocprofv3 kernel trace lists for the dominant kernel."""
        if self.wide:
            sq = 65 <= self.w <= 128 and self.d <= 128          # (square layers of the streaming kernel's widths: cl_wide_head_bwd_supported's range)
            return (("wide_sq_kernel + wide_gemm_kernel" if sq else "wide_gemm_kernel (+ wide_stream_kernel)") +
                    (" (slot likelihood: deterministic stores)" if self.deterministic else ""))
        obs = self.obs.children[0] if isinstance(self.obs, ObsChunks) else self.obs
        ma = self._mlp_args(0, None, None, obs)
        if self.blocks is not None:
            ma = self._block_args(ma, obs, len(self.blocks) - 1)
            if getattr(self, "chain_lane", False):
                ma.dZ0_out = ptr(obs.chain_dact[len(self.blocks) - 2])
            else:
                ma.dX_out = ptr(obs.chain_dact[len(self.blocks) - 2])
        elif self.laue and not obs.fused_laue:
            mode = 2 if mode == 0 else mode
            ma.dO_ext = ptr(obs.laue_dO)
        elif self.peel and mode == 0:
            ma = self._peel_args(ma, obs)
        buf = C.create_string_buffer(128)
        check(min(0, self.lib.cl_mlp_kernel_name(C.byref(ma), mode, buf, 128)), "cl_mlp_kernel_name")
        return buf.value.decode()

    def _w_ll(self, obs: ObsData) -> float:
        """Weight of one log-likelihood term: sum / S, or with `kl_weight` the mean over the S x N terms of the observation set
        the model is called on (reference variational.py:172-177) -- the validation set's own N for `NLL_val` (:257-260)."""
        return 1.0 / self.S if self.model.kl_weight is None else 1.0 / (self.S * obs.N_total)

    def _noise_to_device(self, u_f, eta, obs: Optional[ObsData] = None):
        """Injected noise arrives in the reference's (S, R) / (S, N_total) orientation; device layout is [R][S] / [N][S]."""
        obs = self.obs if obs is None else obs
        du = de = None
        if u_f is not None:
            u = torch.as_tensor(_np(u_f), dtype=torch.float32).reshape(self.S, self.R)
            du = u.t().contiguous().to(self.device)
        if eta is not None:
            e = torch.as_tensor(_np(eta), dtype=torch.float32).reshape(self.S, obs.N_total)
            if obs.rows is not None:
                de = e[:, torch.as_tensor(obs.rows)].t().contiguous().to(self.device)
            else:
                de = e[:, obs.start:obs.start + obs.N].t().contiguous().to(self.device)
        return du, de

    # ------------------------------------------------------------------------------------------------------
    def forward_backward(self, step: int, u_f=None, eta=None, ipred_out=None):
        """Enqueue the loss + gradient part of a step (everything up to, not including, the optimizer)."""
        lib, st = self.lib, _stream()
        tn = self._tn_args(step, u_f)
        tn.kl_part = ptr(self.kl_part)       # the forward launch stores its workgroups' KL sums, the backward launch of this step adds them up
        if self.double_wilson:
            tn.kl_part_dw = ptr(self.kl_part_dw)
        # the step's accumulators are cleared by the forward launch itself (no memset launch in front of it): the flat gradient and
        # the scalars by all its threads, the dz_f rows of the reflections it samples by their threads
        n_own = (self.shard.kl_end - self.shard.kl_begin) if self.owner else self.R
        if self.ws_step.numel() <= max(1 << 20, 64 * n_own * self.S):
            tn.zero_ptr, tn.zero_n, tn.zero_dzf = self.ws_step.data_ptr(), int(self.ws_step.numel()), ptr(self.dz_f)
        else:                                # (per-image layers over many images: a gradient vector far longer than the launch: memset)
            self._zero_step()
        check(lib.cl_tn_forward(C.byref(tn), st), "cl_tn_forward")
        if self.double_wilson:
            check(lib.cl_dw_prior_forward(C.byref(tn), st), "cl_dw_prior_forward")
        self._pending_reduce = None
        dist_on = (self.shard.world > 1 and not getattr(self, "local_only", False)) or getattr(self, "force_allreduce", False)
        # Row split, two-piece message (`model.split_message = True` / CARELESS_HIP_SPLIT_MESSAGE=1; measured in DESIGN 5.2): everything
        # behind a and b in the flat gradient -- the scaler's, the image scales', Ev11's part -- is final once the fused kernel and the
        # partial reduction are through, so its all-reduce starts then (asynchronously, on the communicator's stream) and runs beside
        # cl_tn_backward; a's and b's gradient follows when that kernel is done.  (Not with a trainable double-Wilson r: its gradient
        # sits in the tail and comes out of cl_tn_backward.)
        split = dist_on and not self.owner and not self.dw_trainable and self._want_split_message()
        self._data_term(self.obs, step, eta, ipred_out, st, defer_reduce=not split)
        tail_work = None
        if split:
            import torch.distributed as dist
            tail_work = dist.all_reduce(self.grads[2 * self.R:], op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
        if self._pending_reduce is not None:
            # ... and the reduction of the fused kernel's per-workgroup scaler-gradient partials rides in extra workgroups of the
            # backward launch (independent work: one launch instead of two)
            tn.red_partials, tn.red_nparts, tn.red_P, tn.red_out = self._pending_reduce
            self._pending_reduce = None
        check(lib.cl_tn_backward(C.byref(tn), st), "cl_tn_backward")
        if self.owner:
            # this rank's share of |d a|^2 + |d b|^2 into the message, next to the scaler's gradient
            check(lib.cl_owner_qnorm(ptr(self.grads), self.R, self.shard.kl_begin, self.shard.kl_end, ptr(self.msg_norm),
                                     ptr(self.own_scratch), ptr(self.stop_flag), st), "cl_owner_qnorm")
        if split:
            from careless_amd.distributed import allreduce_flat_
            allreduce_flat_(self.grads[: 2 * self.R], self.process_group)
            tail_work.wait()         # (the current stream waits for the first piece; the host does not block)
        elif dist_on:
            self._allreduce()        # local_only: a test hook that leaves the per-rank partial gradient in place
        self._keep = (u_f, eta, ipred_out)

    def _want_split_message(self) -> bool:
        want = getattr(self.model, "split_message", None)
        if want is None:
            want = os.environ.get("CARELESS_HIP_SPLIT_MESSAGE", "0") == "1"
        return bool(want)

    def _zero_step(self):
        """Clear the step's accumulators: dz_f, the flat gradient, the scalars.  An owner-mode rank only ever touches the dz_f rows and
        the q gradients of its own reflections (those gradients are rewritten, not accumulated, by nobody else): it clears its slice
        of dz_f and everything from the gradients on."""
        if not self.owner:
            self.ws.zero_()
            return
        r0, r1 = self.shard.kl_begin, self.shard.kl_end
        self.dz_f[r0 * self.S:r1 * self.S].zero_()
        self.ws_step.zero_()

    def _det_reduce(self, obs, st):
        det, lay = obs.det, self.layout
        a = DetArgs()
        # (records stored in reflection order -- det_slot -- need no gather list)
        a.dzf_obs, a.perm_refl, a.seg_refl = ptr(det["dzf"]), (None if det["slot"] is not None else ptr(det["refl"][0])), ptr(det["refl"][1])
        a.R, a.S, a.dz_f = self.R, self.S, ptr(self.dz_f)
        if lay.n_img > 0:
            a.dimg_obs, a.perm_img, a.seg_img = ptr(det["dimg"]), ptr(det["img"][0]), ptr(det["img"][1])
            a.n_images, a.d_img = det["M"], self.grads.data_ptr() + 4 * lay.off_img
        a.nll_part, a.nparts, a.scalars = ptr(det["nll"]), int(det["nll"].numel()), ptr(self.scalars)
        a.stop_flag = ptr(self.stop_flag)
        if self.ev11:
            a.ev11_part, a.n_ev11, a.d_ev11 = ptr(det["ev11"]), int(det["ev11"].numel()) // 3, self.grads.data_ptr() + 4 * lay.off_ev11
        check(self.lib.cl_det_reduce(C.byref(a), st), "cl_det_reduce")

    def _data_term(self, obs: ObsData, step: int, eta, ipred_out, st, _piece: bool = False, defer_reduce: bool = False):
        """NLL of `obs` into scalars[NLL] and its gradient into dz_f / the flat gradient (scaler + image scales).  `defer_reduce`:
        a single fused launch leaves its partial reduction to the caller (`self._pending_reduce`: forward_backward hands it to the
        step's cl_tn_backward launch)."""
        lib, lay = self.lib, self.layout
        if getattr(obs, "empty", False):
            return
        if isinstance(obs, ObsChunks):
            for piece in obs.children:
                self._data_term(piece, step, eta, ipred_out, st, _piece=True)
            if self.deterministic:
                self._det_reduce(obs, st)
            return
        if self.scaler_frozen and self.frozen_fast and self._frozen_ok(obs):
            self._data_term_frozen(obs, step, eta, ipred_out, st)
            if self.deterministic and not _piece:
                self._det_reduce(obs, st)
            return
        if self.wide:
            self._data_term_wide(obs, step, eta, ipred_out, st)
            if self.deterministic and not _piece:
                self._det_reduce(obs, st)
            return
        ma = self._mlp_args(step, eta, ipred_out, obs)
        if self.blocks is not None:
            self._data_term_chain(ma, obs, step, eta, ipred_out, st)
            if self.deterministic and not _piece:
                self._det_reduce(obs, st)
            return
        if self.laue and obs.fused_laue:
            # single pass: the harmonic group sums happen inside the fused kernel; the padded slots (no rows, iconv = 0,
            # reference formatter.py:637-640 / laue.py:24) only add their constant -- and, with Ev11, its gradient
            self._fused_step_launch(ma, obs, st)
            self._laue_pad_slots(ma, obs, st)
        elif self.laue:
            self._laue_passes(ma, obs, step, eta, ipred_out, st)
        else:
            # (deterministic mode: every workgroup of the launch STORES its NLL slot, det["grid"] slots per piece -- nothing to clear;
            #  the slots a short last piece leaves unwritten were zero-initialised and are never touched)
            self._fused_step_launch(ma, obs, st)
        if self.peel and not (self.laue and not obs.fused_laue):
            self._peel_reduce(obs, st, ma.n_obs)
        elif defer_reduce and not _piece and not self.deterministic:
            self._pending_reduce = (ptr(obs.partials), obs.grid, lay.P, self.grads.data_ptr() + 4 * lay.off_mlp)
        else:
            check(lib.cl_reduce_partials(ptr(obs.partials), obs.grid, lay.P, self.grads.data_ptr() + 4 * lay.off_mlp,
                                         ptr(self.stop_flag), st), "cl_reduce_partials")
        if self.deterministic and not _piece:
            self._det_reduce(obs, st)

    def _laue_pad_slots(self, ma: MlpArgs, obs: ObsData, st):
        """The padded slots of a packed Laue observation set (no rows, iconv = 0: reference formatter.py:637-640 / laue.py:24) add their
        constant to the NLL -- and, with Ev11, its gradient -- through one small `cl_laue_likelihood` launch."""
        lib = self.lib
        npad = int(obs.pad_iobs.numel())
        if npad <= 0:
            return
        uniform = getattr(obs, "pad_uniform", False)            # all padded slots alike: one slot, weight x their number
        nslot = 1 if uniform else npad
        obs.pad_iconv[: nslot * self.S].zero_()
        la = LaueArgs()
        la.iobs, la.sig, la.iconv = ptr(obs.pad_iobs), ptr(obs.pad_sig), ptr(obs.pad_iconv)
        la.n_obs, la.S = nslot, self.S
        la.lik_kind, la.dof, la.lik_const = self.lik_kind, self.dof, self.lik_const
        la.w_ll = ma.w_ll * (npad if uniform else 1)
        la.scalars, la.stop_flag = ptr(self.scalars), ptr(self.stop_flag)
        la.ev11, la.d_ev11 = ma.ev11, ma.d_ev11
        if self.deterministic:      # the padded slots' workgroups store their NLL behind the fused launch's parts
            det = obs.det_parent.det
            la.nll_part = det["nll"].data_ptr() + 8 * det["pieces"] * det["grid"]
            if self.ev11:
                la.ev11_part = det["ev11"].data_ptr() + 4 * 3 * _lib.CL_EV11_WAVES * det["pieces"] * det["grid"]
        check(lib.cl_laue_likelihood(C.byref(la), st), "cl_laue_likelihood")

    def _frozen_ok(self, obs: ObsData) -> bool:
        """The sampling / likelihood kernels take this observation image as it is: rows in the caller's order (not packed by image or
        harmonic group, not sorted by image), and -- deterministic mode -- rows that are their own slot."""
        if getattr(obs, "host_inputs", None) is None:
            return False
        if obs.fused_laue:
            # harmonic groups in the packed order of the single-pass kernels (round 6): `cl_frozen_rows` in its two-call form -- an engine
            # built around a frozen scaler keeps that layout for Laue data (_build_obs)
            return self._frozen_layout and self.FROZEN_LAUE_PACKED and not self.deterministic and obs.tile_img is None
        if obs.row_map is not None or getattr(obs, "perm", None) is not None:
            return False
        if self.imgl is not None and not self._frozen_layout:
            return False
        return not (self.deterministic and obs.laue)

    def _data_term_frozen(self, obs: ObsData, step: int, eta, ipred_out, st):
        """The data term of a step whose scaling model is frozen (round 5): (loc, sigma) of every row once per `train_model` call -- any
        scaler the package runs, through `scaler_forward` -- then per step only what depends on the sampled amplitudes: sample the scale,
        predict, log-prob and its gradient to dz_f (and the Evans-2011 terms) on the slot kernels of the two-pass path
        (`cl_slot_rows`, or `cl_laue_predict / _likelihood / _backward` for harmonic groups).  The scaler's own gradient is not computed at
        all: the reference takes gradients of `trainable_variables` only (variational.py:201), so its "Grad Norm" does not see it either."""
        if getattr(obs, "locsig_epoch", None) != self._frozen_epoch:
            sl = obs.rows if obs.rows is not None else slice(obs.start, obs.start + obs.N)
            md = _np(BaseModel.get_metadata(obs.host_inputs))
            md = md.reshape(md.shape[0], -1)[sl]
            ids = _np(BaseModel.get_image_id(obs.host_inputs)).reshape(-1)[sl] if self.imgl is not None else None
            obs.laue_loc, obs.laue_sig = scaler_forward(self.mlp, md, self.imgl, ids)
            if getattr(obs, "laue_dO", None) is None:
                obs.laue_dO = torch.empty(obs.N * 2, dtype=torch.float32, device=self.device)
            if not hasattr(obs, "harmonic_id") and not obs.fused_laue:
                obs.harmonic_id = None
            if getattr(obs, "laue_iconv", None) is None and not obs.fused_laue:      # (rows that are their own slot never touch it; the entry point wants a pointer)
                obs.laue_iconv = torch.empty(obs.N * self.S if obs.harmonic_id is not None else 4, dtype=torch.float32, device=self.device)
            if obs.rows is not None and getattr(obs, "row_index", None) is None:
                obs.row_index = torch.as_tensor(np.asarray(obs.rows, dtype=np.int64), device=self.device)     # the noise key of every row
            obs.locsig_epoch = self._frozen_epoch
        ma = self._mlp_args(step, eta, ipred_out, obs)
        if obs.fused_laue:
            self._frozen_laue(ma, obs, step, eta, ipred_out, st)
            return
        keyed = getattr(obs, "row_index", None) is not None and (eta is not None or ipred_out is not None)      # (injected noise on rows that are not a contiguous range: the slot kernels index it by local row)
        if obs.harmonic_id is None and not self.deterministic and self.FROZEN_SORTED_ROWS and not keyed:
            self._frozen_rows(ma, obs, step, eta, ipred_out, st)
            return
        self._slot_likelihood(ma, obs, step, eta, ipred_out, st, frozen=True)

    def _frozen_rows(self, ma: MlpArgs, obs: ObsData, step: int, eta, ipred_out, st):
        """Monochromatic rows behind a frozen scaler (round 6, `cl_frozen_rows`): the rows sorted by reflection ONCE per training -- the
        scaler's output is a constant, so no layout constraint binds their order -- with (loc, sigma), the image scale and the global row
        number (the noise key) of every row beside them; per step one row-per-thread launch that sums the amplitude gradients of a
        reflection's rows inside the wave and stores them (no float atomics into dz_f), plus the small launch for the runs that cross a
        wave border.  The image scales are part of the frozen scaling model: their gradient is not computed either."""
        fz = getattr(obs, "frozen_sorted", None)
        if fz is None or fz["epoch"] != self._frozen_epoch:
            dev = self.device
            rid = obs.refl_id[: obs.N]
            order = torch.argsort(rid, stable=True)
            take = lambda t: t[: obs.N].index_select(0, order).contiguous()
            if ma.use_img:
                img_id = obs.image_id[: obs.N].long()
                scales = torch.cat([torch.ones(1, dtype=torch.float32, device=dev), self.params[self.layout.off_img: self.layout.off_img + self.layout.n_img].detach()])
                aim = scales.index_select(0, img_id).index_select(0, order).contiguous()      # (image 0 is pinned to 1: image.py:23-25)
            else:
                aim = None
            if getattr(obs, "row_index", None) is not None:
                key = obs.row_index[: obs.N].index_select(0, order).to(torch.int32).contiguous()
            else:
                key = (order + int(obs.start)).to(torch.int32).contiguous()
            n_waves = (obs.N + 63) // 64
            fz = obs.frozen_sorted = dict(
                epoch=self._frozen_epoch, refl=take(rid).to(torch.int32), loc=take(obs.laue_loc), sigma=take(obs.laue_sig), aim=aim,
                iobs=take(obs.iobs), sig=take(obs.sig), key=key,
                edge_rid=torch.empty(2 * n_waves, dtype=torch.int32, device=dev),
                edge_val=torch.empty(max(int(self.lib.cl_frozen_edge_floats(obs.N, self.S)), 1), dtype=torch.float32, device=dev))
        fa = FrozenArgs()
        fa.refl_id, fa.loc, fa.sigma, fa.aim = ptr(fz["refl"]), ptr(fz["loc"]), ptr(fz["sigma"]), ptr(fz["aim"])
        fa.iobs, fa.sig, fa.key = ptr(fz["iobs"]), ptr(fz["sig"]), ptr(fz["key"])
        fa.obs_offset, fa.n = int(obs.start), int(obs.N)
        fa.R, fa.S = self.R, self.S
        fa.z_f, fa.dz_f = ptr(self.z_f), ptr(self.dz_f)
        # (several launches into one dz_f -- the pieces of a chunked shard -- and the double-Wilson prior, whose dlog p / dz is in dz_f
        #  before the data term: add with atomics instead of storing)
        fa.accumulate = 1 if (getattr(obs, "is_piece", False) or self.double_wilson) else 0
        fa.lik_kind, fa.dof, fa.lik_const = self.lik_kind, self.dof, self.lik_const
        fa.shift, fa.w_ll = ma.shift, ma.w_ll
        row0 = getattr(obs, "row0", 0)                  # piece of a chunked shard: its rows inside the shard's eta / ipred arrays
        fa.eta = None if eta is None else eta.data_ptr() + 4 * self.S * row0
        fa.seed, fa.step = self.seed, step & 0xFFFFFFFF
        fa.scalars, fa.stop_flag = ptr(self.scalars), ptr(self.stop_flag)
        fa.ipred_out = None if ipred_out is None else ipred_out.data_ptr() + 4 * self.S * row0
        fa.ev11, fa.d_ev11 = ma.ev11, ma.d_ev11
        fa.edge_rid, fa.edge_val = ptr(fz["edge_rid"]), ptr(fz["edge_val"])
        check(self.lib.cl_frozen_rows(C.byref(fa), st), "cl_frozen_rows")

    def _frozen_laue(self, ma: MlpArgs, obs: ObsData, step: int, eta, ipred_out, st):
        """Harmonic groups behind a frozen scaler (round 6): `cl_frozen_rows` twice -- in the packed order of the single-pass kernels the group
        sums, the likelihood and every row's amplitude gradient (stored per row: the rows of a group belong to different reflections), then
        the same rows in reflection order, gathered and summed per reflection like monochromatic rows -- and the padded slots' constant."""
        fz = getattr(obs, "frozen_sorted", None)
        dev = self.device
        if obs.noise_row is not None and (eta is not None or ipred_out is not None):
            raise NotImplementedError("injected noise / ipred_out on a shard of harmonic groups behind a frozen scaler (a parity-test input: "
                                      "set model.frozen_scaler_fast_path = False)")
        if fz is None or fz["epoch"] != self._frozen_epoch:
            rm = obs.row_map.long()
            valid = rm >= 0
            rmc = rm.clamp(min=0)
            loc_p = obs.laue_loc.index_select(0, rmc).contiguous()
            sig_p = obs.laue_sig.index_select(0, rmc).contiguous()
            if ma.use_img:
                scales = torch.cat([torch.ones(1, dtype=torch.float32, device=dev), self.params[self.layout.off_img: self.layout.off_img + self.layout.n_img].detach()])
                aim = scales.index_select(0, obs.image_id.long().clamp(min=0)).contiguous()
            else:
                aim = None
            key = obs.noise_row if obs.noise_row is not None else (rmc + int(obs.start)).to(torch.int32).contiguous()
            active = torch.nonzero(valid & (obs.refl_id >= 0)).flatten()
            order = torch.argsort(obs.refl_id.index_select(0, active), stable=True)
            src = active.index_select(0, order)
            refl_sorted = obs.refl_id.index_select(0, src).to(torch.int32).contiguous()
            n2 = int(src.numel())
            # the first pass writes a row's gradients at its position in reflection order (a 4 S-byte store per row), the second reads them front to back
            dst = torch.full((obs.n_pad,), -1, dtype=torch.int32, device=dev)
            dst[src] = torch.arange(n2, dtype=torch.int32, device=dev)
            fz = obs.frozen_sorted = dict(
                epoch=self._frozen_epoch, loc=loc_p, sigma=sig_p, aim=aim, key=key, dst=dst, refl_sorted=refl_sorted, n2=n2,
                gbuf=torch.zeros(max(n2, 1) * self.S, dtype=torch.float32, device=dev),
                edge_rid=torch.empty(2 * ((n2 + 63) // 64) + 2, dtype=torch.int32, device=dev),
                edge_val=torch.empty(max(int(self.lib.cl_frozen_edge_floats(n2, self.S)), 1), dtype=torch.float32, device=dev))
        row0 = getattr(obs, "row0", 0)
        fa = FrozenArgs()
        fa.refl_id, fa.loc, fa.sigma, fa.aim = ptr(obs.refl_id), ptr(fz["loc"]), ptr(fz["sigma"]), ptr(fz["aim"])
        fa.iobs, fa.sig, fa.key = ptr(obs.iobs), ptr(obs.sig), ptr(fz["key"])
        fa.obs_offset, fa.n = int(obs.start), int(obs.n_pad)
        fa.R, fa.S = self.R, self.S
        fa.z_f, fa.dz_f = ptr(self.z_f), ptr(self.dz_f)
        fa.lik_kind, fa.dof, fa.lik_const = self.lik_kind, self.dof, self.lik_const
        fa.shift, fa.w_ll = ma.shift, ma.w_ll
        fa.eta = None if eta is None else eta.data_ptr() + 4 * self.S * row0
        fa.seed, fa.step = self.seed, step & 0xFFFFFFFF
        fa.scalars, fa.stop_flag = ptr(self.scalars), ptr(self.stop_flag)
        fa.ipred_out = None if ipred_out is None else ipred_out.data_ptr() + 4 * self.S * row0
        fa.ev11, fa.d_ev11 = ma.ev11, ma.d_ev11
        fa.gmeta, fa.gbuf, fa.src = ptr(obs.gmeta), ptr(fz["gbuf"]), ptr(fz["dst"])
        check(self.lib.cl_frozen_rows(C.byref(fa), st), "cl_frozen_rows (harmonic groups)")
        if fz["n2"] > 0:
            fb = FrozenArgs()
            fb.refl_id, fb.gbuf = ptr(fz["refl_sorted"]), ptr(fz["gbuf"])
            fb.n, fb.R, fb.S = fz["n2"], self.R, self.S
            fb.dz_f, fb.stop_flag = ptr(self.dz_f), ptr(self.stop_flag)
            fb.accumulate = 1 if (getattr(obs, "is_piece", False) or self.double_wilson) else 0
            fb.edge_rid, fb.edge_val = ptr(fz["edge_rid"]), ptr(fz["edge_val"])
            check(self.lib.cl_frozen_rows(C.byref(fb), st), "cl_frozen_rows (per-reflection sums)")
        self._laue_pad_slots(ma, obs, st)

    def _peel_bufs(self, obs: ObsData):
        """Buffers of the peeled first layer for one observation set: its pre-activations and dZ_0 (feature-major, like meta_t), the
        weight-gradient partials; per engine: the peeled scaler's parameters and reduced gradient."""
        pb = getattr(obs, "peel", None)
        if pb is None:
            rows = int(self.lib.cl_mlp_meta_rows(self.w))
            nparts = int(self.lib.cl_peel_parts(obs.n_pad))     # (packed layouts: every row of the padded axis may be a real one)
            pb = obs.peel = dict(u=torch.zeros(rows * obs.n_pad, dtype=torch.float32, device=self.device),
                                 dz0=torch.zeros(rows * obs.n_pad, dtype=torch.float32, device=self.device),
                                 parts=torch.empty(nparts * (self.w * self.d + self.w), dtype=torch.float32, device=self.device), nparts=nparts)
        if getattr(self, "peel_params", None) is None:
            Pp = int(self.lib.cl_mlp_param_count(self.w, self.w, self.L))
            self.peel_params = torch.zeros(Pp, dtype=torch.float32, device=self.device)
            self.peel_grad = torch.zeros(Pp, dtype=torch.float32, device=self.device)
        return pb

    def _peel_args(self, ma: MlpArgs, obs: ObsData) -> MlpArgs:
        """The fused launch behind the peeled first layer: the layer's pre-activations as metadata, identity first layer, dZ_0 out."""
        pb = self._peel_bufs(obs)
        a = MlpArgs()
        C.memmove(C.byref(a), C.byref(ma), C.sizeof(MlpArgs))
        a.meta_t, a.d, a.mlp, a.dZ0_out = ptr(pb["u"]), self.w, ptr(self.peel_params), ptr(pb["dz0"])
        return a

    def _fused_step_launch(self, ma: MlpArgs, obs: ObsData, st):
        """cl_elbo_mono_fwd_bwd, behind cl_peel_forward when the first layer is peeled (self.peel)."""
        if not self.peel:
            check(self.lib.cl_elbo_mono_fwd_bwd(C.byref(ma), obs.grid, st), "cl_elbo_mono_fwd_bwd")
            return
        pb, lay = self._peel_bufs(obs), self.layout
        check(self.lib.cl_peel_forward(ptr(obs.meta_t), ma.n_obs, obs.n_pad, self.d, self.w, self.L, self.params.data_ptr() + 4 * lay.off_mlp, ptr(pb["u"]),
                                       ptr(self.peel_params), ptr(self.peel_grad), int(self.peel_grad.numel()), ptr(self.stop_flag), st), "cl_peel_forward")
        a = self._peel_args(ma, obs)
        check(self.lib.cl_elbo_mono_fwd_bwd(C.byref(a), obs.grid, st), "cl_elbo_mono_fwd_bwd")

    def _peel_reduce(self, obs: ObsData, st, n_obs: int):
        """Gradient of a step with a peeled first layer: the launch's partials -> the peeled scaler's gradient; layer 0's own gradient
        from dZ_0 and the metadata, everything behind it copied over (cl_peel_backward)."""
        pb, lay = obs.peel, self.layout
        check(self.lib.cl_reduce_partials(ptr(obs.partials), obs.grid, int(self.peel_grad.numel()), ptr(self.peel_grad), ptr(self.stop_flag), st),
              "cl_reduce_partials")
        check(self.lib.cl_peel_backward(ptr(obs.meta_t), n_obs, obs.n_pad, self.d, self.w, self.L, ptr(pb["dz0"]), ptr(self.peel_grad),
                                        self.grads.data_ptr() + 4 * lay.off_mlp, ptr(pb["parts"]), pb["nparts"], ptr(self.stop_flag), st), "cl_peel_backward")

    def _block_args(self, ma: MlpArgs, obs: ObsData, k: int) -> MlpArgs:
        """Arguments of block k of the chain: its slice of the parameters, its input (metadata or the previous block's output)."""
        lay, b = self.layout, self.blocks[k]
        a = MlpArgs()
        C.memmove(C.byref(a), C.byref(ma), C.sizeof(MlpArgs))
        a.mlp = self.params.data_ptr() + 4 * (lay.off_mlp + b.off)
        a.d, a.L = b.d_in, b.l1 - b.l0
        a.meta_t = ptr(obs.meta_t) if k == 0 else ptr(obs.chain_act[k - 1])
        return a

    def _data_term_chain(self, ma: MlpArgs, obs: ObsData, step: int, eta, ipred_out, st):
        """Scaler deeper than one launch: forward through the head-less blocks (activations to HBM), the last block does the
        likelihood and its own backward (recomputing its forward), then the blocks are walked back, each recomputing its
        forward from its stored input.  8 P_mm flops per observation instead of 6, plus 2 x 8 w bytes per block boundary."""
        lib, lay, K = self.lib, self.layout, len(self.blocks)
        gptr = lambda k: self.grads.data_ptr() + 4 * (lay.off_mlp + self.blocks[k].off)
        for k in range(K - 1):
            a = self._block_args(ma, obs, k)
            a.act_out = ptr(obs.chain_act[k])
            check(lib.cl_mlp_forward(C.byref(a), obs.grid, st), "cl_mlp_forward")
        a = self._block_args(ma, obs, K - 1)
        if getattr(self, "chain_lane", False):
            # the last block on the lane kernel: dZ_0 of its first layer out (into a buffer of the boundary's shape), then dX = W_0^T dZ_0
            if getattr(obs, "chain_dz0", None) is None:
                obs.chain_dz0 = torch.zeros_like(obs.chain_dact[K - 2])
            a.dZ0_out = ptr(obs.chain_dz0)
            check(lib.cl_elbo_mono_fwd_bwd(C.byref(a), obs.grid, st), "cl_elbo_mono_fwd_bwd")
            check(lib.cl_chain_dx(ptr(obs.chain_dz0), a.mlp, a.n_obs, obs.n_pad, self.w, self.w, ptr(obs.chain_dact[K - 2]), ptr(self.stop_flag), st), "cl_chain_dx")
        else:
            a.dX_out = ptr(obs.chain_dact[K - 2])
            if self.laue:
                self._laue_passes(a, obs, step, eta, ipred_out, st)
            else:
                check(lib.cl_elbo_mono_fwd_bwd(C.byref(a), obs.grid, st), "cl_elbo_mono_fwd_bwd")
        check(lib.cl_reduce_partials(ptr(obs.partials), obs.grid, self.blocks[K - 1].P, gptr(K - 1), ptr(self.stop_flag), st),
              "cl_reduce_partials")
        for k in range(K - 2, -1, -1):
            a = self._block_args(ma, obs, k)
            a.dH_ext = ptr(obs.chain_dact[k])
            a.dX_out = ptr(obs.chain_dact[k - 1]) if k > 0 else None
            check(lib.cl_mlp_backward_ext(C.byref(a), obs.grid, st), "cl_mlp_backward_ext")
            check(lib.cl_reduce_partials(ptr(obs.partials), obs.grid, self.blocks[k].P, gptr(k), ptr(self.stop_flag), st),
                  "cl_reduce_partials")

    def evaluate_nll(self, obs: ObsData, key: int, u_f=None, eta=None) -> float:
        """NLL of another observation set under the current parameters with fresh Monte-Carlo noise -- what
        `model.test_on_batch(validation_data)` reports as "NLL" (reference variational.py:257-260).  `u_f` (S, R) / `eta`
        (S, N_val) inject the noise (parity tests).  Uses the step workspace (call it between steps); synchronises."""
        lib, st = self.lib, _stream()
        u_f, eta = self._noise_to_device(u_f, eta, obs)
        self._zero_step()
        tn = self._tn_args(key, u_f)
        check(lib.cl_tn_forward(C.byref(tn), st), "cl_tn_forward")
        self._data_term(obs, key, eta, None, st)
        if self.owner and self.shard.world > 1 and not getattr(self, "local_only", False):
            # an owner-mode rank holds the validation rows of ITS reflections (make_obs): the set's NLL is the sum over the ranks
            import torch.distributed as dist
            t = self.scalars[0:1].clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.process_group)
            return float(t.item())
        torch.cuda.synchronize()
        return float(self.scalars[0].item())

    def make_obs(self, inputs) -> ObsData:
        """Device image of another observation set (validation data) for `evaluate_nll`."""
        if BaseModel.is_laue(inputs) != self.laue:
            raise ValueError("validation data and training data differ in kind (mono / Laue)")
        rows = None
        if self.owner:
            # validation rows follow their reflections' owner (only the owner samples them); a rank none of whose reflections occurs
            # in the set launches nothing and contributes 0 to the sum
            rid = _np(BaseModel.get_refl_id(inputs)).reshape(-1)
            rows = np.nonzero((rid >= self.shard.kl_begin) & (rid < self.shard.kl_end))[0]
            if len(rows) == 0:
                return _EmptyObs(int(len(rid)))
        o = self._build_obs(inputs, 0, None, None, None, rows=rows)
        if o.d != self.d:
            raise ValueError("validation metadata width differs from the training data")
        if self.blocks is not None:
            o.alloc_chain(self.lib, self.blocks, self.w, self.device)
        return o

    def _laue_passes(self, ma: MlpArgs, obs: ObsData, step: int, eta, ipred_out, st):
        """Harmonic deconvolution (reference likelihoods/laue.py:9-47): scaler forward, predict + group sums, likelihood on
        the slots, gradient broadcast back to the rows, scaler backward from dL/d(loc, sigma)."""
        lib = self.lib
        ma.loc_out, ma.sig_out = ptr(obs.laue_loc), ptr(obs.laue_sig)
        check(lib.cl_mlp_forward(C.byref(ma), obs.grid, st), "cl_mlp_forward")
        self._slot_likelihood(ma, obs, step, eta, ipred_out, st)
        ma.dO_ext = ptr(obs.laue_dO)
        check(lib.cl_mlp_backward_ext(C.byref(ma), obs.grid, st), "cl_mlp_backward_ext")

    def _slot_args(self, ma: MlpArgs, obs: ObsData, step: int, eta, ipred_out, a: int = 0, n: Optional[int] = None) -> LaueArgs:
        """Arguments of the slot likelihood kernels for the rows [a, a + n) of `obs` (default: all of them)."""
        la = LaueArgs()
        n = obs.N - a if n is None else n
        off = lambda t, size: None if t is None else t.data_ptr() + size * a
        la.refl_id, la.image_id, la.harmonic_id = off(obs.refl_id, 4), off(obs.image_id, 4), off(obs.harmonic_id, 4)
        la.loc, la.sigma, la.iobs, la.sig = off(obs.laue_loc, 4), off(obs.laue_sig, 4), off(obs.iobs, 4), off(obs.sig, 4)
        la.n_obs, la.obs_offset = n, obs.start + a
        la.img, la.use_img = ma.img, ma.use_img
        la.z_f, la.R, la.S = ptr(self.z_f), self.R, self.S
        la.lik_kind, la.dof, la.lik_const = self.lik_kind, self.dof, self.lik_const
        la.shift, la.w_ll = ma.shift, ma.w_ll
        la.eta = off(eta, 4 * self.S)
        la.seed, la.step = self.seed, step & 0xFFFFFFFF
        la.iconv, la.dz_f, la.d_img, la.dO = off(obs.laue_iconv, 4 * self.S), ptr(self.dz_f), ma.d_img, off(obs.laue_dO, 8)
        la.scalars, la.ipred_out, la.stop_flag = ptr(self.scalars), off(ipred_out, 4 * self.S), ptr(self.stop_flag)
        la.ev11, la.d_ev11 = ma.ev11, ma.d_ev11
        la.row_index = off(obs.row_index, 8)
        return la

    def _slot_likelihood(self, ma: MlpArgs, obs: ObsData, step: int, eta, ipred_out, st, frozen: bool = False):
        """From (loc, sigma) per row in obs.laue_loc / laue_sig: sample, predict, group sums, slot likelihood (NLL into the
        scalars), its gradient back on the rows -> dz_f, d(image scales), obs.laue_dO = dL/d(loc, sigma) per row."""
        lib = self.lib
        if obs.harmonic_id is not None:
            obs.laue_iconv.zero_()          # (group sums accumulate by atomics; rows that are their own slot -- harmonic_id None -- store)
        la = self._slot_args(ma, obs, step, eta, ipred_out)
        if self.deterministic and obs.harmonic_id is None:
            # no float atomics: the amplitude gradient of every (row, sample) and the image-scale term of every row are stored (records in
            # reflection order: det_slot), every workgroup stores its NLL; cl_det_reduce sums them in a fixed order after the backward pass
            det = obs.det_parent.det
            la.dzf_obs, la.dimg_obs, la.det_slot = ptr(det["dzf"]), ptr(det["dimg"]), ptr(det["slot"])
            la.nll_part = det["nll"].data_ptr() + 8 * det["pieces"] * det["grid"]
            if self.ev11:
                la.ev11_part = det["ev11"].data_ptr() + 4 * 3 * _lib.CL_EV11_WAVES * det["pieces"] * det["grid"]
            check(lib.cl_slot_rows(C.byref(la), st), "cl_slot_rows")
            return
        if obs.harmonic_id is None and self.SLOT_ROWS_ONE_LAUNCH:
            # every row its own slot (monochromatic data on the layer-by-layer path): one launch, no round trip through iconv
            check(lib.cl_slot_rows(C.byref(la), st), "cl_slot_rows")
            return
        check(lib.cl_laue_predict(C.byref(la), st), "cl_laue_predict")
        check(lib.cl_laue_likelihood(C.byref(la), st), "cl_laue_likelihood")
        if frozen:
            la.dO = None                # nobody takes dL/d(loc, sigma) of a frozen scaler: amplitude gradients only (no clearing, no row sums)
        check(lib.cl_laue_backward(C.byref(la), st), "cl_laue_backward")

    def _allreduce(self):
        from careless_amd.distributed import allreduce_flat_
        # row split: the whole flat gradient (2 R + P + ... floats); reflection-owner split: the scaler's part and the norm terms
        allreduce_flat_(self.msg if self.owner else self.grads, self.process_group)

    def sync_owned(self):
        """Reflection-owner mode: every rank has updated a and b of its own reflections only; after training all ranks need all of
        them (the output step, saved weights, a later `train_model` call).  One sum all-reduce of a vector that is zero outside the
        rank's own ranges -- once per training run, not per step.  Adam's moments stay with the owner (ownership is fixed for the
        engine's lifetime)."""
        if not self.owner or self.shard.world <= 1 or getattr(self, "local_only", False):
            return
        import torch.distributed as dist
        if not dist.is_initialized():
            return
        from careless_amd.distributed import gather_owned_
        gather_owned_(self.params, self.R, self.shard.kl_begin, self.shard.kl_end, self.process_group)

    def optimizer_step(self, step_index: int):
        lib, st, opt = self.lib, _stream(), self.opt
        n = self.layout.n
        clipnorm = float(opt.clipnorm or 0.0)
        use_seg = clipnorm > 0.0
        # the clip modes that need the norm BEFORE the update get their own pass; otherwise the norm rides inside the Adam kernel
        norm_first = use_seg or float(opt.global_clipnorm or 0.0) > 0.0
        if norm_first:
            check(lib.cl_grad_sqnorm(ptr(self.grads), n, ptr(self.seg_off), self.nseg, ptr(self.seg_sq) if use_seg else None,
                                     ptr(self.scalars), ptr(self.frozen) if self.any_frozen else None, ptr(self.stop_flag), st), "cl_grad_sqnorm")
            if self.owner:
                # the pass above saw this rank's own q gradients (the others' entries are zero here) and the all-reduced tail: trade
                # the own share (cl_owner_qnorm's double accumulators) for the sum over the ranks that came back in the message
                tot, own = self.msg_norm.double(), self.own_scratch
                self.scalars[2] += tot[0] - own[0]
                self.scalars[3] += tot[1] - own[1]
                if use_seg:
                    self.seg_sq[0] += tot[2] - own[2]
                    self.seg_sq[1] += tot[3] - own[3]
        self.t += 1
        t = self.t
        a = AdamArgs()
        a.p = ptr(self.params); a.g = ptr(self.grads); a.m = ptr(self.adam_m); a.v = ptr(self.adam_v)
        a.n = n
        a.alpha = opt.learning_rate * math.sqrt(1.0 - opt.beta_2 ** t) / (1.0 - opt.beta_1 ** t)
        a.beta1, a.beta2, a.adam_eps = opt.beta_1, opt.beta_2, opt.epsilon
        # tf_keras applies the FIRST active clip mode only -- clipnorm, else global_clipnorm, else clipvalue
        # (`_BaseOptimizer._clip_gradients` [3P-recall]; oracle.clip_grads): the kernel sees one mode
        a.global_clipnorm = 0.0 if clipnorm > 0.0 else float(opt.global_clipnorm or 0.0)
        a.clipnorm = clipnorm
        a.clipvalue = 0.0 if (clipnorm > 0.0 or a.global_clipnorm > 0.0) else float(opt.clipvalue or 0.0)
        a.seg_off, a.nseg = ptr(self.seg_off), self.nseg
        a.seg_sq = ptr(self.seg_sq)
        a.frozen = ptr(self.frozen) if self.any_frozen else None
        a.scalars = ptr(self.scalars)
        a.stop_flag = ptr(self.stop_flag)
        a.norm_out = None if norm_first else ptr(self.scalars)
        if self.owner:
            # update what this rank owns -- a and b of its reflections -- and the replicated tail; the norm fused into the call covers
            # the tail, the q part over all ranks arrives in the message
            r0, r1, R = self.shard.kl_begin, self.shard.kl_end, self.R
            a.n_ranges = 3
            a.range_begin[0], a.range_end[0] = r0, r1
            a.range_begin[1], a.range_end[1] = R + r0, R + r1
            a.range_begin[2], a.range_end[2] = 2 * R, n
            a.norm_skip_ranges = 2
            a.norm_extra = None if norm_first else ptr(self.msg_norm)
        n_part = 0
        if not norm_first:
            # the norm fused into the update leaves one pair of sums per workgroup; cl_step_finalize adds them up
            a.norm_part = ptr(self.norm_part)
            n_part = int(lib.cl_adam_grid(C.byref(a)))
        check(lib.cl_adam_step(C.byref(a), st), "cl_adam_step")
        check(lib.cl_step_finalize(ptr(self.scalars), self.kl_mult, ptr(self.history_buf), step_index,
                                   ptr(self.stop_flag), ptr(self.norm_part) if n_part else None, n_part, st), "cl_step_finalize")

    def alloc_history(self, steps: int):
        self.history_buf = torch.zeros(max(1, steps) * _lib.CL_HIST_STRIDE, dtype=torch.float64, device=self.device)
        self.stop_flag.zero_()
        self._hist_reduced = False
        self.rdw_hist = (torch.zeros(max(1, steps), self.layout.n_dwr, dtype=torch.float32, device=self.device)
                         if self.dw_trainable else None)

    def train_step(self, step_index: int, u_f=None, eta=None):
        """One full ELBO step; `step_index` indexes the history buffer, the noise key is the optimizer iteration."""
        if self.dw_trainable:           # the "rDW_i" metrics are the values used in this step's forward pass (wilson.py:173-174)
            lay = self.layout
            self.rdw_hist[step_index] = torch.sigmoid(self.params[lay.off_dwr:lay.off_dwr + lay.n_dwr])
        self.forward_backward(self.t, u_f, eta)
        self.optimizer_step(step_index)

    def read_history(self, steps: int) -> Dict[str, List[float]]:
        """Synchronise and convert the device history to the reference's dict of lists (variational.py:262-268).
        Steps after the first non-finite gradient norm were skipped on the device and are dropped, which reproduces
        the reference's early `break` (:271-274)."""
        if self.shard.world > 1 and not getattr(self, "local_only", False) and not self._hist_reduced:
            self._hist_reduced = True
            # the records hold this rank's partial NLL / KL: summed over the ranks once, in fp64 (careless_amd/distributed.py)
            from careless_amd.distributed import allreduce_history_
            allreduce_history_(self.history_buf, _lib.CL_HIST_STRIDE, self.kl_mult, self.process_group)
        h = self.history_buf.view(-1, _lib.CL_HIST_STRIDE)[:steps].cpu().numpy()
        keep = h[:, 4] == 0.0
        h = h[keep]
        out = {"Grad Norm": h[:, 3].tolist()}
        out["loss"] = h[:, 0].tolist()
        out["F KLDiv"] = h[:, 1].tolist()
        out["NLL"] = h[:, 2].tolist()
        if self.dw_trainable:
            r = self.rdw_hist[:steps].cpu().numpy()[keep]
            for i in range(r.shape[1]):
                out[f"rDW_{i}"] = r[:, i].tolist()
        return out

    # -- accessors used by tests -----------------------------------------------------------------------------
    def loss_terms(self) -> Dict[str, float]:
        s = self.scalars.cpu().numpy()
        return {"nll": float(s[0]), "kl": float(s[1]), "loss": float(s[0] + self.kl_mult * s[1])}

    def grad_tensors(self) -> List[torch.Tensor]:
        """Gradients split per trainable tensor, in the oracle's order and Keras shapes
        [q_loc_raw, q_scale_raw, W_0 (in,out), b_0, ..., W_o, b_o, image scales, per-image layers, Ev11, double-Wilson r]."""
        return self._split(self.grads)

    def param_tensors(self) -> List[torch.Tensor]:
        """The trainable tensors themselves, same order and shapes as `grad_tensors`."""
        return self._split(self.params)

    def _split(self, g: torch.Tensor) -> List[torch.Tensor]:
        lay = self.layout
        out = [g[0:self.R], g[self.R:2 * self.R]]
        base = lay.off_mlp
        for off, o, i, boff in self.mlp.layer_slices(self.d):
            out.append(g[base + off: base + off + o * i].view(o, i).t())
            out.append(g[base + boff: base + boff + o])
        if lay.n_img > 0:
            out.append(g[lay.off_img: lay.off_img + lay.n_img])
        if lay.n_imgl > 0:
            K, M, w = self.imgl.n_image_layers, self.imgl.max_images, self.w
            for k in range(K):
                o = lay.off_imgl + k * M * (w * w + w)
                out.append(g[o: o + M * w * w].view(M, w, w))
                out.append(g[o + M * w * w: o + M * (w * w + w)].view(M, w))
        if lay.n_ev11 > 0:
            out.append(g[lay.off_ev11: lay.off_ev11 + lay.n_ev11])
        if lay.n_dwr > 0:
            out.append(g[lay.off_dwr: lay.off_dwr + lay.n_dwr])
        return out


# ------------------------------------------------------------------------------------------------------------
# stand-alone helpers behind the plugin protocol methods
# ------------------------------------------------------------------------------------------------------------
def tn_sample(q, n: int, seed=None, u_f=None) -> torch.Tensor:
    """`TruncatedNormal.sample(n)` -> (n, R) via `cl_tn_forward` (Philox noise unless `u_f` (n,R) is injected)."""
    dev = require_gpu("TruncatedNormal.sample")
    lib = _lib.get_lib()
    R = q.loc_raw.numel()
    loc_raw = q.loc_raw.to(dev, torch.float32).contiguous()
    scale_raw = q.scale_raw.to(dev, torch.float32).contiguous()
    low = q.low.to(dev, torch.float32).contiguous()
    centric = torch.zeros(R, dtype=torch.uint8, device=dev)
    es = torch.ones(R, dtype=torch.float32, device=dev)
    z = torch.empty(R * n, dtype=torch.float32, device=dev)
    sc = torch.zeros(_lib.CL_SC_COUNT, dtype=torch.float64, device=dev)
    du = None
    if u_f is not None:
        du = torch.as_tensor(_np(u_f), dtype=torch.float32).reshape(n, R).t().contiguous().to(dev)
    a = TnArgs()
    a.q_loc_raw, a.q_scale_raw, a.low, a.centric, a.es = ptr(loc_raw), ptr(scale_raw), ptr(low), ptr(centric), ptr(es)
    a.R, a.S = R, n
    a.high, a.eps = q.high, q.scale_shift
    a.w_kl, a.kl_grad_mult = 0.0, 0.0
    a.kl_begin, a.kl_end = 0, 0
    a.u_f = ptr(du)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    a.seed, a.step = int(seed) & 0xFFFFFFFFFFFFFFFF, 0
    a.z_f, a.scalars = ptr(z), ptr(sc)
    check(lib.cl_tn_forward(C.byref(a), _stream()), "cl_tn_forward")
    return z.view(R, n).t()


def tn_moments(q, high_m4=np.inf, want=("mean", "std", "m4")):
    """Moments of the truncated-normal posterior for the output step via `cl_tn_moments` (reference surrogate_posteriors.py:23-27,
    55-102; consumed at io/manager.py:188-197): dict with `mean`, `std` (fp32 torch tensors on the device) and `m4` (fp64).
    `high_m4` is the upper bound of the fourth moment (the reference's `moment_4(high=np.inf)` default); mean / std use q.high."""
    dev = require_gpu("TruncatedNormal moments")
    lib = _lib.get_lib()
    R = q.loc_raw.numel()
    loc_raw = q.loc_raw.to(dev, torch.float32).contiguous()
    scale_raw = q.scale_raw.to(dev, torch.float32).contiguous()
    low = q.low.to(dev, torch.float32).contiguous()
    out = {}
    if "mean" in want:
        out["mean"] = torch.empty(R, dtype=torch.float32, device=dev)
    if "std" in want:
        out["std"] = torch.empty(R, dtype=torch.float32, device=dev)
    if "m4" in want:
        out["m4"] = torch.empty(R, dtype=torch.float64, device=dev)
    hi4 = float("inf") if high_m4 is None else float(high_m4)
    check(lib.cl_tn_moments(ptr(loc_raw), ptr(scale_raw), ptr(low), R, float(q.high), hi4, float(q.scale_shift),
                            ptr(out.get("mean")), ptr(out.get("std")), ptr(out.get("m4")), _stream()), "cl_tn_moments")
    return out


def predict_moments(scale_mean, scale_std, refl_id, mom):
    """E[I] and var[I] of every observation (fp64 numpy arrays) via `cl_predict_moments` (reference variational.py:80-121) from the scale's
    moments per row (device tensors), the rows' reflection ids and `tn_moments(q)`."""
    dev = require_gpu("prediction_mean_stddev")
    lib = _lib.get_lib()
    sm = scale_mean.to(dev, torch.float32).contiguous().reshape(-1)
    ss = scale_std.to(dev, torch.float32).contiguous().reshape(-1)
    n = int(sm.numel())
    rid = torch.as_tensor(_np(refl_id).reshape(-1).astype(np.int32), device=dev)
    if int(rid.numel()) != n or int(ss.numel()) != n:
        raise ValueError("scale moments and refl_id differ in length")
    iexp = torch.empty(n, dtype=torch.float64, device=dev)
    ivar = torch.empty(n, dtype=torch.float64, device=dev)
    check(lib.cl_predict_moments(ptr(sm), ptr(ss), ptr(rid), n, ptr(mom["mean"]), ptr(mom["std"]), ptr(mom["m4"]), int(mom["mean"].numel()),
                                 ptr(iexp), ptr(ivar), _stream()), "cl_predict_moments")
    return iexp.cpu().numpy(), ivar.cpu().numpy()


def scaler_forward(mlp, metadata, imgl=None, image_id=None):
    """loc, sigma of the scaler's Normal for every row of `metadata` via `cl_mlp_forward`; `imgl` + `image_id` add the
    per-image layers of a `NeuralImageScaler`."""
    dev = require_gpu("MLPScaler.call")
    lib = _lib.get_lib()
    md = _np(metadata).astype(np.float32)
    md = md.reshape(md.shape[0], -1) if md.ndim > 1 else md.reshape(-1, 1)
    N, d = md.shape
    mlp.build(d)
    if mlp.flat.device != dev:
        mlp.flat = mlp.flat.to(dev)
    too_deep = imgl is not None and mlp.n_layers + imgl.n_image_layers > int(lib.cl_mlp_max_layers_imgl(mlp.width))
    if mlp.width > 64 or d > 64 or too_deep:
        # wider (or, with per-image layers, deeper) than the fused kernel holds: the layer-by-layer GEMM kernels (see
        # ElboEngine._data_term_wide), forward only; per-image layers run grouped on the rows sorted by image
        w, L, st = mlp.width, mlp.n_layers, _stream()
        order = seg = None
        keep = []
        if imgl is not None:
            imgl.build(d)
            if imgl.flat.device != dev:
                imgl.flat = imgl.flat.to(dev)
            ids = _np(image_id).reshape(-1).astype(np.int64)
            if ids.size != N or (ids.size and (ids.min() < 0 or ids.max() >= imgl.max_images)):
                raise ValueError("image_id does not match the metadata / exceeds max_images")
            order = np.argsort(ids, kind="stable")
            md = md[order]
            seg = np.concatenate([[0], np.cumsum(np.bincount(ids, minlength=imgl.max_images))]).astype(np.int64)
        ld0, ldw = int(lib.cl_wide_ld(d)), int(lib.cl_wide_ld(w))
        chunk = max(128, min(N, ((256 << 20) // (8 * max(ld0, ldw))) // 128 * 128))
        if seg is None:
            chunks = [(a, min(N, a + chunk), 0, None) for a in range(0, N, chunk)]
        else:                                   # whole images per chunk
            chunks, m0, M = [], 0, imgl.max_images
            while m0 < M:
                m1 = m0 + 1
                while m1 < M and seg[m1 + 1] - seg[m0] <= chunk:
                    m1 += 1
                if seg[m1] > seg[m0]:
                    chunks.append((int(seg[m0]), int(seg[m1]), m0, torch.as_tensor((seg[m0:m1 + 1] - seg[m0]).astype(np.int32), device=dev)))
                m0 = m1
            chunk = max(b - a for a, b, _, _ in chunks)
        loc = torch.empty(N, dtype=torch.float32, device=dev)
        sig = torch.empty(N, dtype=torch.float32, device=dev)
        x = torch.zeros(chunk * ld0, dtype=torch.float32, device=dev)
        hb = [torch.zeros(chunk * ldw, dtype=torch.float32, device=dev) for _ in range(2)]
        bij = _lib.CL_BIJ_EXP if mlp.scale_bijector == "exp" else _lib.CL_BIJ_SOFTPLUS
        base = mlp.flat.data_ptr()
        for a, b, m0, sg in chunks:
            x.view(chunk, ld0)[: b - a, :d] = torch.as_tensor(md[a:b], device=dev)
            src, off, fan_in, nl = (x.data_ptr(), ld0), 0, d, 0
            for l in range(L):
                dst = hb[nl & 1]
                check(lib.cl_wide_dense_forward(src[0], src[1], base + 4 * off, base + 4 * (off + w * fan_in), b - a, fan_in, w, mlp.leakiness, 1,
                                                ptr(dst), ldw, None, st), "cl_wide_dense_forward")
                src, off, fan_in, nl = (dst.data_ptr(), ldw), off + w * fan_in + w, w, nl + 1
            for k in range(imgl.n_image_layers if imgl is not None else 0):
                dst = hb[nl & 1]
                M = imgl.max_images
                kb = imgl.flat.data_ptr() + 4 * k * M * (w * w + w)
                if w > 128:                         # wider than the grouped streaming kernel holds: the tiled kernel over a list of row pieces
                    from careless_amd.wide import image_tiles
                    tl = image_tiles(sg.cpu().numpy(), dev)
                    keep.append(tl)
                    check(lib.cl_wide_image_forward_tiles(src[0], src[1], kb + 4 * m0 * w * w, kb + 4 * (M * w * w + m0 * w), ptr(sg), ptr(tl), tl.numel() // 2,
                                                          w, mlp.leakiness, ptr(dst), ldw, None, st), "cl_wide_image_forward_tiles")
                else:
                    check(lib.cl_wide_image_forward(src[0], src[1], kb + 4 * m0 * w * w, kb + 4 * (M * w * w + m0 * w), ptr(sg), sg.numel() - 1, b - a, w,
                                                    mlp.leakiness, ptr(dst), ldw, None, st), "cl_wide_image_forward")
                src, nl = (dst.data_ptr(), ldw), nl + 1
            check(lib.cl_wide_head_forward(src[0], src[1], base + 4 * off, b - a, w, bij, mlp.epsilon, loc.data_ptr() + 4 * a,
                                           sig.data_ptr() + 4 * a, None, st), "cl_wide_head_forward")
        if order is not None:                   # back to the caller's row order
            inv = torch.as_tensor(order, device=dev)
            loc_c, sig_c = torch.empty_like(loc), torch.empty_like(sig)
            loc_c[inv], sig_c[inv] = loc, sig
            return loc_c, sig_c
        return loc, sig
    keep = []
    if imgl is not None:
        imgl.build(d)
        if imgl.flat.device != dev:
            imgl.flat = imgl.flat.to(dev)
        ids = _np(image_id).reshape(-1).astype(np.int64)
        if ids.size != N or (ids.size and (ids.min() < 0 or ids.max() >= imgl.max_images)):
            raise ValueError("image_id does not match the metadata / exceeds max_images")
        pos, n_pad, tile_img, row_map = pack_by_image(ids)
        meta_t = np.zeros((int(lib.cl_mlp_meta_rows(d)), n_pad), dtype=np.float32)
        meta_t[:d, pos] = md.T
        keep = [torch.as_tensor(tile_img, device=dev), torch.as_tensor(row_map, device=dev)]
    else:
        n_pad = ((N + TILE - 1) // TILE) * TILE
        meta_t = np.zeros((int(lib.cl_mlp_meta_rows(d)), n_pad), dtype=np.float32)
        meta_t[:d, :N] = md.T
    meta_t = torch.as_tensor(meta_t, device=dev)
    loc = torch.empty(N, dtype=torch.float32, device=dev)
    sig = torch.empty(N, dtype=torch.float32, device=dev)
    a = MlpArgs()
    a.meta_t, a.n_obs, a.n_pad = ptr(meta_t), N, n_pad
    a.mlp = ptr(mlp.flat)
    a.d, a.w, a.L, a.leak = d, mlp.width, mlp.n_layers, mlp.leakiness
    a.bij_kind = _lib.CL_BIJ_EXP if mlp.scale_bijector == "exp" else _lib.CL_BIJ_SOFTPLUS
    a.eps = mlp.epsilon
    a.S, a.R = 1, 1
    a.loc_out, a.sig_out = ptr(loc), ptr(sig)
    if imgl is not None:
        a.n_obs = n_pad
        a.imgl, a.n_imgl, a.n_images = ptr(imgl.flat), imgl.n_image_layers, imgl.max_images
        a.tile_img, a.row_map = ptr(keep[0]), ptr(keep[1])
    grid = min(max(1, int(lib.cl_mlp_default_grid())), n_pad // TILE)
    max_plain = int(lib.cl_mlp_max_layers(mlp.width))
    if imgl is None and mlp.n_layers > max_plain:
        # deeper than one launch: chain of layer blocks (see ElboEngine._data_term_chain)
        blocks = chain_plan(d, mlp.width, mlp.n_layers, max_plain)
        rows = int(lib.cl_mlp_meta_rows(mlp.width))
        x = meta_t
        for b in blocks:
            a.mlp = mlp.flat.data_ptr() + 4 * b.off
            a.d, a.L, a.meta_t = b.d_in, b.l1 - b.l0, ptr(x)
            if not b.final:
                y = torch.zeros(rows, n_pad, dtype=torch.float32, device=dev)
                a.act_out = ptr(y)
                check(lib.cl_mlp_forward(C.byref(a), grid, _stream()), "cl_mlp_forward")
                keep.append(x)
                x = y
            else:
                a.act_out = None
                check(lib.cl_mlp_forward(C.byref(a), grid, _stream()), "cl_mlp_forward")
        return loc, sig
    check(lib.cl_mlp_forward(C.byref(a), grid, _stream()), "cl_mlp_forward")
    return loc, sig


def debug_noise(seed: int, step: int, S: int, n: int, offset: int = 0, kind: int = 1) -> torch.Tensor:
    """The noise the kernels draw for (seed, step): kind 0 = q(F) uniforms [n][S], kind 1 = scale normals [n][S]."""
    dev = require_gpu("debug_noise")
    out = torch.empty(n * S, dtype=torch.float32, device=dev)
    check(_lib.get_lib().cl_debug_noise(int(seed) & 0xFFFFFFFFFFFFFFFF, step, S, n, offset, kind, ptr(out), _stream()),
          "cl_debug_noise")
    return out.view(n, S)

