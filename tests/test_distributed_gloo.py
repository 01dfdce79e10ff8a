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


def _owner_worker(rank, world, port, q):
    from careless_amd.distributed import allreduce_flat_, gather_owned_
    from careless_amd.engine import owner_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    kw = dict(N=203, R=21, d0=5, L=2, w=16, S=3)
    data, cfg, params, x, u_f, eta = util.make_problem(**kw)
    R = kw["R"]
    sh = owner_shard(np.asarray(data["refl_id"]), R, rank, world)
    rows = torch.as_tensor(sh.rows)
    xs = O.ElboInputs(refl_id=x.refl_id[rows], image_id=x.image_id[rows], metadata=x.metadata[rows], iobs=x.iobs[rows],
                      sigiobs=x.sigiobs[rows], centric=x.centric, multiplicity=x.multiplicity, low=x.low, sigma=x.sigma)
    kl_mask = torch.zeros(R, dtype=torch.bool)
    kl_mask[sh.kl_begin:sh.kl_end] = True
    u, e = torch.as_tensor(u_f, dtype=torch.float64), torch.as_tensor(eta[:, sh.rows], dtype=torch.float64)
    out, grads = O.elbo_value_and_grads(params, xs, cfg, u, e, kl_mask=kl_mask)
    flat = torch.cat([g.reshape(-1) for g in grads]).to(torch.float32)
    own = torch.zeros(2 * R, dtype=torch.bool)
    own[sh.kl_begin:sh.kl_end] = True
    own[R + sh.kl_begin:R + sh.kl_end] = True
    foreign = float(flat[: 2 * R][~own].abs().max())             # nothing of another rank's reflections in this rank's gradient
    # the step's message: the replicated tail and this rank's share of |d a|^2 + |d b|^2 (engine.ElboEngine.msg)
    msg = torch.cat([flat[2 * R:], (flat[: 2 * R].double() ** 2).sum().float().reshape(1)])
    allreduce_flat_(msg)
    # after "training": a parameter vector of which the rank changed its own entries only
    p = torch.arange(2 * R + 5, dtype=torch.float32)
    p[: 2 * R][own] += 100.0 * (rank + 1)
    gather_owned_(p, R, sh.kl_begin, sh.kl_end)
    q.put((rank, sh.kl_begin, sh.kl_end, sh.rows, flat[: 2 * R].numpy(), msg.numpy(), foreign, p.numpy(), float(out["nll"]), float(out["kl"])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reflection_owner_shards_sum_to_full_batch():
    """The reflection-owner split on CPU (gloo, world size 2; the ORACLE stands in for the kernels): the ranks' q gradients are
    disjoint and concatenate to the full-batch one, the all-reduced message holds the full scaler gradient and the q gradient's
    squared norm, and `gather_owned_` leaves every rank with every owner's parameters."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_owner_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kw = dict(N=203, R=21, d0=5, L=2, w=16, S=3)
    R = kw["R"]
    data, cfg, params, x, u_f, eta = util.make_problem(**kw)
    out, grads = O.elbo_value_and_grads(params, x, cfg, torch.as_tensor(u_f, dtype=torch.float64),
                                        torch.as_tensor(eta, dtype=torch.float64))
    full = torch.cat([g.reshape(-1) for g in grads]).numpy()
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == R
    rows = np.concatenate([g[3] for g in got])
    assert len(rows) == kw["N"] and len(np.unique(rows)) == kw["N"]
    qsum = got[0][4] + got[1][4]
    tol = dict(rtol=2e-5, atol=1e-6 * np.abs(full).max())
    assert np.allclose(qsum, full[: 2 * R], **tol)
    for g in got:
        assert g[6] == 0.0
        assert np.allclose(g[5][:-1], full[2 * R:], **tol)
        assert np.isclose(g[5][-1], (full[: 2 * R].astype(np.float64) ** 2).sum(), rtol=1e-5)
        want = np.arange(2 * R + 5, dtype=np.float32)
        for o in got:
            want[o[1]:o[2]] += 100.0 * (o[0] + 1)
            want[R + o[1]:R + o[2]] += 100.0 * (o[0] + 1)
        assert np.array_equal(g[7], want)
    assert np.isclose(got[0][8] + got[1][8], float(out["nll"]), rtol=1e-10) and np.isclose(got[0][9] + got[1][9], float(out["kl"]), rtol=1e-10)


def test_owner_bounds_balance_observations_and_fall_back_when_impossible():
    from careless_amd.engine import owner_bounds, owner_shard
    rng = np.random.default_rng(3)
    rid = rng.integers(0, 5000, size=160_000)
    for world in (2, 3, 4, 8):
        b = owner_bounds(rid, 5000, world)
        assert b[0] == 0 and b[-1] == 5000 and np.all(np.diff(b) > 0)
        cnt = [int(((rid >= b[r]) & (rid < b[r + 1])).sum()) for r in range(world)]
        assert sum(cnt) == len(rid) and max(cnt) - min(cnt) <= 4 * np.bincount(rid).max()          # balanced to a few reflections' worth
        sh = [owner_shard(rid, 5000, r, world) for r in range(world)]
        assert all(s.owner and s.start == 0 and s.stop == len(s.rows) for s in sh)
        assert all(np.all(np.diff(s.rows) > 0) for s in sh)                                       # ascending: the caller's order is kept
        assert all(np.all((rid[s.rows] >= s.kl_begin) & (rid[s.rows] < s.kl_end)) for s in sh)
    # reflections without observations are owned too (their KL term and update are somebody's); skewed counts still balance
    rid = np.concatenate([np.zeros(1000, dtype=np.int64), rng.integers(10, 20, size=1000)])
    b = owner_bounds(rid, 40, 2)
    assert b.tolist() == [0, 1, 40] or (b[0] == 0 and b[-1] == 40)
    assert owner_bounds(np.array([0, 0, 0, 1]), 5, 3) is None and owner_shard(np.array([0, 0, 0, 1]), 5, 0, 3) is None   # -> row split


def _format_worker(rank, world, port, q, fail):
    """`careless._format_once_per_node` under two gloo ranks: only rank 0 formats; every rank ends up with the same arrays (read-only maps of
    rank 0's files) and the same ASU object; a formatting error on rank 0 raises on EVERY rank (no rank is left in a barrier)."""
    import careless_amd.careless as cc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for k in ("LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        os.environ.pop(k, None)
    if fail == "two_nodes":                      # one rank per "node": each formats its own copy (advisor, round 4: a second node used to hang)
        os.environ.update(LOCAL_RANK="0", LOCAL_WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"n": 0}
    if fail == "load" and rank == 1:             # a rank-local failure AFTER the formatting flag: the others must not be left in a barrier
        def broken_load(*a, **k):
            raise OSError("/dev/shm is full")
        cc.np.load = broken_load

    def fake_format(parser):
        calls["n"] += 1
        if fail is True:
            raise OSError("cannot read the reflection file")
        rng = np.random.default_rng(3)
        inputs = (rng.integers(0, 9, (50, 1)), rng.integers(0, 4, (50, 1)), np.zeros((50, 1), np.int64), rng.normal(size=(50, 3)).astype(np.float32),
                  rng.normal(size=(50, 1)).astype(np.float32), np.ones((50, 1), np.float32))
        return inputs, {"asu": "collection", "n": 9}
    cc._format = fake_format
    try:
        inputs, rac = cc._format_once_per_node(None, rank, world)
        ok = len(inputs) == 6 and rac == {"asu": "collection", "n": 9} and (rank == 0 or not inputs[3].flags.writeable)
        q.put((rank, "ok" if ok else "bad", calls["n"], float(np.asarray(inputs[3], dtype=np.float64).sum())))
    except OSError as e:
        q.put((rank, "raised", calls["n"], str(e)))
    except RuntimeError as e:
        q.put((rank, "raised", calls["n"], str(e)))
    if dist.is_initialized():
        dist.destroy_process_group()


def test_reflection_files_are_formatted_once_per_node_and_failures_reach_every_rank():
    world = 2
    ctx = mp.get_context("spawn")
    for fail in (False, True, "load", "two_nodes"):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_format_worker, args=(r, world, port, q, fail)) for r in range(world)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=180) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert [g[2] for g in got] == ([1, 1] if fail == "two_nodes" else [1, 0])     # one formatting rank per node
        if fail in (True, "load"):
            assert [g[1] for g in got] == ["raised", "raised"]     # both ranks leave with an error; nobody waits in a collective
        else:
            assert [g[1] for g in got] == ["ok", "ok"] and got[0][3] == got[1][3]
        assert not [f for f in os.listdir("/dev/shm") if f.startswith(f"careless_amd_127.0.0.1_{port}")]    # nothing left behind
