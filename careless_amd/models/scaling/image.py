"""Per-image scale factors and per-image layers.

Mirror of `careless/models/scaling/image.py` (reference): `ImageScaler` (:9-42; image 0 pinned to 1, the other M-1
scales trainable, initial value 1), `HybridImageScaler` (:44-63; MLP distribution scaled by the gathered image scale) and
`NeuralImageScaler` (:66-125; `--image-layers`: the Dense stack is followed by layers whose kernel and bias belong to the
image of the observation).  Gathers, products, per-image matrix products and all gradients happen inside the fused HIP kernel.
"""
from __future__ import annotations

import numpy as np
import torch

from careless_amd.models.scaling.base import Scaler
from careless_amd.models.scaling.nn import NormalDistribution


def _to_index(x, device):
    x = x.detach() if hasattr(x, "detach") else torch.as_tensor(np.asarray(x))
    return x.reshape(-1).to(device=device, dtype=torch.int64)


class ImageScaler(Scaler):
    """Simple linear image scales (reference image.py:9-42)."""

    def __init__(self, max_images):
        super().__init__()
        self.max_images = int(max_images)
        self._scales = torch.ones(self.max_images - 1, dtype=torch.float32)
        self.trainable = True

    @property
    def scales(self):
        one = torch.ones(1, dtype=self._scales.dtype, device=self._scales.device)
        return torch.cat([one, self._scales])

    @property
    def trainable_variables(self):
        return [self._scales] if self.trainable else []

    def call(self, inputs):
        image_ids = _to_index(self.get_image_id(inputs), self._scales.device)
        return self.scales[image_ids]


class HybridImageScaler(Scaler):
    """A scaler that combines an `ImageScaler` with an `MLPScaler` (reference image.py:44-63)."""

    def __init__(self, mlp_scaler, image_scaler):
        super().__init__()
        self.mlp_scaler = mlp_scaler
        self.image_scaler = image_scaler

    @property
    def trainable(self):
        return self.mlp_scaler.trainable

    @trainable.setter
    def trainable(self, value):
        self.mlp_scaler.trainable = bool(value)
        self.image_scaler.trainable = bool(value)

    @property
    def trainable_variables(self):
        return self.mlp_scaler.trainable_variables + self.image_scaler.trainable_variables

    def save_weights(self, path):
        torch.save({"mlp": {"flat": self.mlp_scaler.flat.detach().cpu(), "input_dim": self.mlp_scaler.input_dim},
                    "image": self.image_scaler._scales.detach().cpu()}, path)

    def load_weights(self, path):
        st = torch.load(path)
        self.mlp_scaler.build(int(st["mlp"]["input_dim"]),
                              device=None if self.mlp_scaler.flat is None else self.mlp_scaler.flat.device)
        self.mlp_scaler.flat.copy_(st["mlp"]["flat"])
        self.image_scaler._scales.copy_(st["image"])

    def call(self, inputs):
        q = self.mlp_scaler(inputs)
        a = self.image_scaler(inputs).to(q.loc.device)
        return NormalDistribution(q.loc, q.scale, shift=q.shift, multiplier=a)


class NeuralImageScaler(Scaler):
    """`MetadataScaler` whose network is followed by `image_layers` per-image layers
    h <- LeakyReLU(W[image_id] h + b[image_id]), W: (max_images, width, width) identity-initialised, b zero
    (reference image.py:66-125).  `flat` holds, per image layer, the kernels [M][out][in] then the biases [M][out]
    (the `cl_mlp_args.imgl` layout of include/careless_hip.h)."""

    def __init__(self, image_layers, max_images, mlp_layers, mlp_width, leakiness=0.01, epsilon=1e-7, scale_bijector=None,
                 scale_multiplier=None):
        super().__init__()
        from careless_amd.models.scaling.nn import MetadataScaler
        self.n_image_layers = int(image_layers)
        if self.n_image_layers < 1:
            raise ValueError("NeuralImageScaler needs at least one image layer")
        self.max_images = int(max_images)
        self.metadata_scaler = MetadataScaler(mlp_layers, mlp_width, leakiness, epsilon=epsilon, scale_bijector=scale_bijector,
                                              scale_multiplier=scale_multiplier)
        if self.metadata_scaler.n_layers < 1:
            raise NotImplementedError("NeuralImageScaler without Dense layers is not supported by the HIP engine")
        self.flat = None

    @property
    def width(self):
        return self.metadata_scaler.width

    def build(self, d: int, device=None):
        self.metadata_scaler.build(d, device=device)
        if self.flat is not None:
            return
        K, M, w = self.n_image_layers, self.max_images, self.width
        flat = torch.zeros(K * M * (w * w + w), dtype=torch.float32)
        for k in range(K):
            o = k * M * (w * w + w)
            flat[o:o + M * w * w] = torch.eye(w).repeat(M, 1, 1).reshape(-1)      # tf.eye(w, w, (M,)) (image.py:73-74)
        self.flat = flat.to(device) if device is not None else flat

    @property
    def image_weights(self):
        """[kernel_0 (M, w, w), bias_0 (M, w), kernel_1, ...] views into the flat buffer (kernel[m] maps in -> out as W h)."""
        K, M, w = self.n_image_layers, self.max_images, self.width
        out = []
        for k in range(K):
            o = k * M * (w * w + w)
            out.append(self.flat[o:o + M * w * w].view(M, w, w))
            out.append(self.flat[o + M * w * w:o + M * (w * w + w)].view(M, w))
        return out

    @property
    def trainable(self):
        return self.metadata_scaler.trainable

    @trainable.setter
    def trainable(self, value):
        self.metadata_scaler.trainable = bool(value)

    @property
    def trainable_variables(self):
        if not self.trainable or self.flat is None:
            return []
        return self.metadata_scaler.weights + self.image_weights

    def save_weights(self, path):
        torch.save({"mlp": {"flat": self.metadata_scaler.flat.detach().cpu(), "input_dim": self.metadata_scaler.input_dim},
                    "image_layers": self.flat.detach().cpu()}, path)

    def load_weights(self, path):
        st = torch.load(path)
        self.build(int(st["mlp"]["input_dim"]), device=None if self.flat is None else self.flat.device)
        self.metadata_scaler.flat.copy_(st["mlp"]["flat"])
        self.flat.copy_(st["image_layers"])

    def call(self, inputs):
        from careless_amd.engine import scaler_forward
        ms = self.metadata_scaler
        loc, sig = scaler_forward(ms, self.get_metadata(inputs), imgl=self, image_id=self.get_image_id(inputs))
        return NormalDistribution(loc, sig, shift=ms.scale_multiplier or 0.0)
