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
    from careless_amd.distributed import allreduce_flat_
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
    n = flat.numel()
    ext = torch.cat([flat, torch.zeros(4)])
    scalars = torch.tensor([float(out["nll"]), float(out["kl"]), 0.0, 0.0], dtype=torch.float64)
    allreduce_flat_(ext, scalars, n)
    if rank == 0:
        q.put((ext[:n].numpy(), scalars.numpy()))
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
    got, scalars = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kw = dict(N=203, R=21, d0=5, L=2, w=16, S=3)
    data, cfg, params, x, u_f, eta = util.make_problem(**kw)
    out, grads = O.elbo_value_and_grads(params, x, cfg, torch.as_tensor(u_f, dtype=torch.float64),
                                        torch.as_tensor(eta, dtype=torch.float64))
    full = torch.cat([g.reshape(-1) for g in grads]).numpy()
    assert np.allclose(got, full, rtol=2e-5, atol=1e-6 * np.abs(full).max())
    assert np.isclose(scalars[0], float(out["nll"]), rtol=1e-6)
    assert np.isclose(scalars[1], float(out["kl"]), rtol=1e-6)
