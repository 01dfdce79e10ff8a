"""Optimizer configuration holder.

The reference builds `tfk.optimizers.Adam(lr, beta_1, beta_2, clipnorm=, clipvalue=, global_clipnorm=)`
(careless/io/manager.py:494-501) and hands it to `model.compile`.  Here `Adam` only carries the hyper-parameters; the
update itself is the fused HIP kernel `cl_adam_step` (careless_amd/csrc/elbo_elem.hip).  Defaults follow tf_keras
(epsilon 1e-7) and the careless CLI (args/optimizer.py: lr 1e-3, beta_1 0.9, beta_2 0.99).
"""
from __future__ import annotations


class Adam:
    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.99, epsilon=1e-7, clipnorm=None, clipvalue=None,
                 global_clipnorm=None):
        if clipnorm is not None and global_clipnorm is not None:
            raise ValueError("At most one of `clipnorm` and `global_clipnorm` can be set")   # tf_keras behaviour
        self.learning_rate = float(learning_rate)
        self.beta_1 = float(beta_1)
        self.beta_2 = float(beta_2)
        self.epsilon = float(epsilon)
        self.clipnorm = clipnorm
        self.clipvalue = clipvalue
        self.global_clipnorm = global_clipnorm
