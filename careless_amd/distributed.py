"""Data-parallel plumbing: one process per GPU, observations sharded, ONE sum all-reduce of the flat gradient per step
(RCCL over xGMI on the GPU box: torch.distributed backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all (no collective call sites); this is the exchange step
the sharded path needs and nothing more: `loss = sum_i l_i / S + sum_h kl_h / S` is additive over observations, so
gradients add, and the KL over reflections is owned by exactly one rank per reflection (`Shard.kl_begin/kl_end`).
"""
from __future__ import annotations

import torch


def allreduce_flat_(grads_ext: torch.Tensor, scalars: torch.Tensor, n: int, group=None) -> None:
    """In-place sum over ranks of the flat gradient `grads_ext[:n]`; the two fp64 loss scalars (NLL, KL) ride in the
    fp32 tail `grads_ext[n:n+2]` of the same message, so a step costs exactly one collective."""
    import torch.distributed as dist
    grads_ext[n:n + 2] = scalars[:2].to(grads_ext.dtype)
    dist.all_reduce(grads_ext, op=dist.ReduceOp.SUM, group=group)
    scalars[:2] = grads_ext[n:n + 2].to(scalars.dtype)
