"""Per-image scale factors.

Mirror of `careless/models/scaling/image.py:9-63` (reference): `ImageScaler` (image 0 pinned to 1, the other M-1
scales trainable, initial value 1) and `HybridImageScaler` (MLP distribution scaled by the gathered image scale).
The gather, the product and the scatter-add of the gradient happen inside the fused HIP kernel.
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
