"""Data-parallel plumbing: one process per GPU, observations sharded, ONE sum all-reduce per step
(RCCL over xGMI on the GPU box: torch.distributed backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).

Two splits (careless_amd/engine.py: make_shard, owner_shard).  Row split: contiguous rows per rank, the message is the whole flat
gradient (2 R + P + ... floats).  Reflection-owner split (monochromatic data, Wilson prior): a rank takes a range of reflections
and every observation of theirs, so everything about q(F_h) of an owned reflection is local and the message is the scaler's
gradient plus four norm terms; `gather_owned_` hands every rank the owners' parameters once, after training.

The reference has no distributed code at all (no collective call sites); this is the exchange step
the sharded path needs and nothing more: `loss = sum_i l_i / S + sum_h kl_h / S` is additive over observations, so
gradients add, and the KL over reflections is owned by exactly one rank per reflection (`Shard.kl_begin/kl_end`).

The logged loss terms (NLL, KL) are NOT part of the per-step message: nothing in the update depends on them, so every rank
keeps its own fp64 partial sums in its device history and the histories are summed ONCE, in fp64, when the host reads them
(`allreduce_history_`).  The multi-GPU history is therefore accumulated in fp64 exactly like the single-GPU one.
"""
from __future__ import annotations

import torch


def allreduce_flat_(grads: torch.Tensor, group=None) -> None:
    """In-place sum over ranks of the flat gradient: the step's one collective."""
    import torch.distributed as dist
    dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=group)


def allreduce_history_(history: torch.Tensor, stride: int, kl_mult: float, group=None) -> None:
    """`history` = fp64 [steps][stride] records (loss, KL, NLL, grad norm, skipped) holding this rank's partial NLL / KL:
    sum those two columns over the ranks and rebuild the loss column.  The gradient norm and the skipped flag are already
    identical on every rank (both derive from the all-reduced gradient)."""
    import torch.distributed as dist
    h = history.view(-1, stride)
    part = h[:, 1:3].contiguous()
    dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    h[:, 1:3] = part
    h[:, 0] = torch.where(h[:, 4] == 0.0, part[:, 1] + float(kl_mult) * part[:, 0], h[:, 0])


def gather_owned_(q_params: torch.Tensor, R: int, r0: int, r1: int, group=None) -> None:
    """Reflection-owner mode, once per training run: `q_params` = [a (R) | b (R)] of which this rank has updated the entries of its
    own reflections [r0, r1) only.  In place, every rank ends up with every owner's values: a sum all-reduce of a copy that is zero
    outside the rank's own ranges (the ranges partition [0, R), so the sum is a concatenation -- exact, no rounding)."""
    import torch.distributed as dist
    full = torch.zeros_like(q_params[: 2 * R])
    full[r0:r1] = q_params[r0:r1]
    full[R + r0:R + r1] = q_params[R + r0:R + r1]
    dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
    q_params[: 2 * R] = full


def check_world(n_obs: int, world: int, n_units: int = None, what: str = "observations") -> None:
    """Every rank must own at least one unit of work, or it would sit out the kernels and meet the others only in the
    collective.  The test depends on global sizes only, so all ranks raise together (nobody is left waiting in RCCL)."""
    n_units = n_obs if n_units is None else n_units
    if world > n_units:
        raise ValueError(f"cannot shard {n_units} {what} over {world} ranks: every rank needs at least one")
