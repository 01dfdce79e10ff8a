"""Neural-network scaling model q(Sigma | metadata).

Mirror of `careless/models/scaling/nn.py:10-120` (reference): `MetadataScaler` / `MLPScaler` = L x Dense(width,
LeakyReLU(0.01), identity kernel init, zero bias) -> Dense(2) -> Normal(loc, bijector(raw) + epsilon).
The parameters live in ONE flat float32 tensor in the device "W^T layout" of include/careless_hip.h (every Dense
kernel stored transposed, then its bias); the Keras-shaped `(in, out)` kernels are exposed as views.  Forward and
backward run in the fused HIP kernel (careless_amd/csrc/elbo_mlp.hip).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from careless_amd.models.scaling.base import Scaler


def _parse_bijector(scale_bijector) -> str:
    """The reference passes tfb.Chain([Shift(eps), Exp()|Softplus()]) (io/manager.py:450-463); here the chain is named.
    None = the class default of nn.py:14-18 (softplus)."""
    if scale_bijector is None:
        return "softplus"
    name = str(scale_bijector).lower()
    if name in ("exp", "softplus"):
        return name
    raise ValueError(f"Unsupported scale bijector type, {scale_bijector}")


class NormalDistribution:
    """What `scaler(inputs)` returns: Normal(loc, scale) optionally shifted (tfb.Shift, nn.py:84-87) and scaled by the
    per-observation image scale (tfb.Scale, image.py:60-63).  Tensors stay on the device of the scaler."""

    def __init__(self, loc: torch.Tensor, scale: torch.Tensor, shift: float = 0.0, multiplier: Optional[torch.Tensor] = None):
        self.loc, self.scale, self.shift, self.multiplier = loc, scale, shift, multiplier

    def mean(self):
        m = self.loc + self.shift
        return m if self.multiplier is None else self.multiplier * m

    def stddev(self):
        return self.scale if self.multiplier is None else self.multiplier.abs() * self.scale

    def sample(self, sample_shape=(), generator=None):
        shape = (sample_shape,) if isinstance(sample_shape, int) else tuple(sample_shape)
        eta = torch.randn(shape + tuple(self.loc.shape), device=self.loc.device, dtype=self.loc.dtype, generator=generator)
        z = self.loc + self.scale * eta + self.shift
        return z if self.multiplier is None else self.multiplier * z


class MetadataScaler(Scaler):
    """Neural network based scaler with simple dense layers; outputs a normal distribution (reference nn.py:27-103)."""

    def __init__(self, n_layers, width, leakiness=0.01, epsilon=1e-7, scale_bijector=None, scale_multiplier=None):
        super().__init__()
        if leakiness is None:
            leakiness = 0.0          # plain ReLU (reference nn.py:57-58)
        self.n_layers = int(n_layers)
        self.width = int(width)
        self.leakiness = float(leakiness)
        self.epsilon = float(epsilon)
        self.scale_bijector = _parse_bijector(scale_bijector)
        self.scale_multiplier = None if scale_multiplier is None else float(scale_multiplier)   # applied with tfb.Shift
        self.input_dim: Optional[int] = None
        self.flat: Optional[torch.Tensor] = None     # W^T layout
        self.trainable = True

    # -- parameters ---------------------------------------------------------------------------------------
    def param_count(self, d: int) -> int:
        w, L = self.width, self.n_layers
        return w * d + w + (L - 1) * (w * w + w) + 2 * w + 2

    def layer_slices(self, d: Optional[int] = None):
        """[(kernel_offset, out, in, bias_offset)] per Dense layer in the flat W^T layout."""
        d = self.input_dim if d is None else d
        w, L = self.width, self.n_layers
        out, off, fan_in = [], 0, d
        for _ in range(L):
            out.append((off, w, fan_in, off + w * fan_in))
            off += w * fan_in + w
            fan_in = w
        out.append((off, 2, fan_in, off + 2 * fan_in))
        return out

    def build(self, d: int, device=None):
        """Identity-initialised kernels (tf.eye(in, out), also when non-square) and zero biases (nn.py:62-78)."""
        if self.flat is not None:
            if d != self.input_dim:
                raise ValueError(f"scaler was built for metadata width {self.input_dim}, got {d}")
            return
        self.input_dim = int(d)
        flat = torch.zeros(self.param_count(d), dtype=torch.float32)
        for off, o, i, _ in self.layer_slices(d):
            flat[off:off + o * i] = torch.eye(o, i).reshape(-1)        # (eye(in,out))^T == eye(out,in)
        self.flat = flat.to(device) if device is not None else flat

    @property
    def weights(self) -> List[torch.Tensor]:
        """[kernel_0 (in,out), bias_0, kernel_1, ...] -- Keras order, views into the flat buffer."""
        out = []
        for off, o, i, boff in self.layer_slices():
            out.append(self.flat[off:off + o * i].view(o, i).t())
            out.append(self.flat[boff:boff + o])
        return out

    def set_weights(self, weights):
        for dst, src in zip(self.weights, weights):
            dst.copy_(torch.as_tensor(np.asarray(src), dtype=torch.float32))

    @property
    def trainable_variables(self):
        return self.weights if self.trainable else []

    def save_weights(self, path):
        torch.save({"flat": self.flat.detach().cpu(), "input_dim": self.input_dim, "n_layers": self.n_layers,
                    "width": self.width}, path)

    def load_weights(self, path):
        st = torch.load(path)
        self.build(int(st["input_dim"]), device=None if self.flat is None else self.flat.device)
        self.flat.copy_(st["flat"])

    # -- forward ------------------------------------------------------------------------------------------
    def call(self, metadata):
        from careless_amd.engine import scaler_forward
        loc, sig = scaler_forward(self, metadata)
        return NormalDistribution(loc, sig, shift=self.scale_multiplier or 0.0)


class MLPScaler(MetadataScaler):
    def call(self, inputs):
        return super().call(self.get_metadata(inputs))
