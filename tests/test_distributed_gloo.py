"""The data-parallel path on CPU: world_size 2, gloo backend.  Checks the product's sharding (`make_shard`) and its one
collective (`allreduce_flat_`): per-rank losses and gradients computed on the shard (by the ORACLE, as the stand-in for the
GPU kernels, which cannot run here) must sum to the full-batch loss and gradient, with the KL counted exactly once."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import elbo_oracle as O
from tests import util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    from careless_amd.distributed import allreduce_flat_, allreduce_history_
    from careless_amd._lib import CL_HIST_STRIDE
    from careless_amd.engine import make_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    kw = dict(N=203, R=21, d0=5, L=2, w=16, S=3)
    data, cfg, params, x, u_f, eta = util.make_problem(**kw)
    sh = make_shard(kw["N"], kw["R"], rank, world)
    sl = slice(sh.start, sh.stop)
    xs = O.ElboInputs(refl_id=x.refl_id[sl], image_id=x.image_id[sl], metadata=x.metadata[sl], iobs=x.iobs[sl],
                      sigiobs=x.sigiobs[sl], centric=x.centric, multiplicity=x.multiplicity, low=x.low, sigma=x.sigma)
    kl_mask = torch.zeros(kw["R"], dtype=torch.bool)
    kl_mask[sh.kl_begin:sh.kl_end] = True
    u, e = torch.as_tensor(u_f, dtype=torch.float64), torch.as_tensor(eta[:, sl], dtype=torch.float64)
    out, grads = O.elbo_value_and_grads(params, xs, cfg, u, e, kl_mask=kl_mask)
    flat = torch.cat([g.reshape(-1) for g in grads]).to(torch.float32)
    allreduce_flat_(flat)
    # the loss terms travel once, in fp64, with the history (three steps' worth of records here; the second one "skipped")
    stride = CL_HIST_STRIDE
    hist = torch.zeros(3 * stride, dtype=torch.float64)
    for i in range(3):
        hist[i * stride + 1], hist[i * stride + 2] = float(out["kl"]) * (i + 1), float(out["nll"]) * (i + 1)
    hist[stride + 4] = 1.0
    allreduce_history_(hist, stride, 0.5)
    if rank == 0:
        q.put((flat.numpy(), hist.view(3, stride).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_sum_to_full_batch():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, hist = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kw = dict(N=203, R=21, d0=5, L=2, w=16, S=3)
    data, cfg, params, x, u_f, eta = util.make_problem(**kw)
    out, grads = O.elbo_value_and_grads(params, x, cfg, torch.as_tensor(u_f, dtype=torch.float64),
                                        torch.as_tensor(eta, dtype=torch.float64))
    full = torch.cat([g.reshape(-1) for g in grads]).numpy()
    assert np.allclose(got, full, rtol=2e-5, atol=1e-6 * np.abs(full).max())
    nll, kl = float(out["nll"]), float(out["kl"])
    for i in (0, 2):                                  # fp64 end to end: the shard sums reproduce the full batch to 1e-12
        assert np.isclose(hist[i, 2], nll * (i + 1), rtol=1e-12) and np.isclose(hist[i, 1], kl * (i + 1), rtol=1e-12)
        assert np.isclose(hist[i, 0], (nll + 0.5 * kl) * (i + 1), rtol=1e-12)
    assert hist[1, 0] == 0.0 and hist[1, 4] == 1.0      # a skipped record keeps its (zero) loss; the flag column is not summed


def _share_worker(rank, world, port, share_dir, q):
    from careless_amd.workloads import make_workload
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the host side of the multi-GPU bench: rank 0 generates, every rank maps the files (no GPU: the model objects are descriptions)
    model, inputs, data, spec = make_workload("dw_50M_normal_5x64_S1", N=20_000, rank=rank, world=world, share_dir=share_dir,
                                              barrier=dist.barrier)
    mapped = all(isinstance(data[k], np.memmap) for k in ("refl_id", "metadata", "iobs", "parent_ids"))
    views = all(np.shares_memory(a, data[k]) for a, k in zip((inputs[0], inputs[3], inputs[4]), ("refl_id", "metadata", "iobs")))
    q.put((rank, mapped, views, int(np.asarray(inputs[0]).sum()), float(np.asarray(inputs[3], dtype=np.float64).sum()), spec["R"], spec["d"]))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_share_one_generated_problem_through_memory_maps(tmp_path):
    """bench.py --gpus N: only rank 0 runs the synthetic generator; every rank maps its files read-only and builds the `inputs`
    tuple as views of the maps (careless_amd/workloads.py: make_workload(share_dir=...)), so a node holds one copy of the problem."""
    from careless_amd.workloads import make_workload
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    share = str(tmp_path / "share")
    procs = [ctx.Process(target=_share_worker, args=(r, world, port, share, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, inputs, data, spec = make_workload("dw_50M_normal_5x64_S1", N=20_000)
    want = (int(np.asarray(inputs[0]).sum()), float(np.asarray(inputs[3], dtype=np.float64).sum()), spec["R"], spec["d"])
    for rank, mapped, views, *rest in got:
        assert mapped and views and tuple(rest) == want, (rank, mapped, views, rest, want)


def test_make_shard_never_hands_out_an_empty_range():
    """Every rank of a data-parallel job owns at least one observation (an idle rank would meet the others only inside the
    collective); impossible splits raise on every rank alike."""
    import pytest
    from careless_amd.engine import laue_group_shard, make_shard
    for n, world in [(10, 4), (9, 4), (5, 4), (4, 4), (1000003, 8), (17, 16)]:
        sh = [make_shard(n, 7, r, world) for r in range(world)]
        assert sh[0].start == 0 and sh[-1].stop == n and all(a.stop == b.start for a, b in zip(sh, sh[1:]))
        assert all(s.stop > s.start for s in sh)
        assert sh[0].kl_begin == 0 and sh[-1].kl_end == 7 and all(a.kl_end == b.kl_begin for a, b in zip(sh, sh[1:]))
    with pytest.raises(ValueError):
        make_shard(3, 7, 0, 4)
    hid = np.repeat(np.arange(5), [3, 1, 2, 1, 1])
    parts = [laue_group_shard(hid, r, 2) for r in range(2)]
    assert parts[0][0] == 0 and parts[0][1] == parts[1][0] and parts[1][1] == 5
    with pytest.raises(ValueError):
        laue_group_shard(np.zeros(8, dtype=np.int64), 0, 2)          # one group cannot be split over two ranks
