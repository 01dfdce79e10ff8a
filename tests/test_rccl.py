"""RCCL in the driver-run suite (round 5).  The multi-GPU step's collectives -- process-group start-up on the `nccl` backend (= RCCL on
ROCm), the per-step all-reduce of the flat gradient, the history all-reduce, the tear-down -- are exercised here on whatever the box
has: with ONE GPU as a world of one (`bench.py --force-dist`, `CARELESS_FORCE_DIST=1` for the command line: every call of the
multi-rank step is made, RCCL has nobody to talk to), each in a FRESH child process as a launcher would start it; with two or more
GPUs also as a real two-rank job in both splits of the observations, compared with the one-rank run.  The reference pins one GPU and
has no counterpart (careless/parser.py:26-40)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env(**kw):
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "CARELESS_FORCE_DIST", "CARELESS_DIST_BACKEND", "CARELESS_HIP_OWNER_SHARD"):
        env.pop(k, None)
    env.update(kw)
    return env


def _bench(argv, env, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_step_over_rccl_in_a_world_of_one():
    """`bench.py --force-dist`: the `nccl` process group comes up in a fresh process, every step all-reduces its flat gradient through
    RCCL, the line reports the backend, and the loss history equals the plain one-rank run's (same in-kernel noise; a sum over one rank
    changes nothing)."""
    argv = ["--nobs", "500000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    d = _bench(argv + ["--force-dist"], _env())
    assert d["backend"] == "nccl" and d["ranks_seen"] == 1 and d["n_gpus"] == 1
    assert d["config"]["loss_finite"] and np.all(np.isfinite(d["loss_history"])) and len(d["loss_history"]) == 4
    plain = _bench(argv, _env())
    assert plain["backend"] is None
    assert np.allclose(d["loss_history"], plain["loss_history"], rtol=1e-5)


def test_command_line_over_rccl_in_a_world_of_one(tmp_path):
    """`python -m careless_amd mono` under WORLD_SIZE=1 CARELESS_FORCE_DIST=1 CARELESS_DIST_BACKEND=nccl: process group on RCCL, the step's
    all-reduce, the output step, the tear-down -- and the same merged amplitudes as the plain run."""
    from careless_amd.io.mtz import read_mtz
    from tests.mtz_fixture import PYP
    flags = "mono --iterations=20 --disable-progress-bar --mlp-layers 3 --test-fraction 0.2 dHKL,image_id".split()
    outs = {}
    for name, env in (("plain", _env()), ("rccl", _env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", CARELESS_FORCE_DIST="1", CARELESS_DIST_BACKEND="nccl"))):
        out = str(tmp_path / name)
        r = subprocess.run([sys.executable, "-m", "careless_amd"] + flags + [PYP, out], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = out
    a, b = read_mtz(outs["plain"] + "_0.mtz"), read_mtz(outs["rccl"] + "_0.mtz")
    assert np.array_equal(a.hkl(), b.hkl()) and np.allclose(a.columns["F"], b.columns["F"], rtol=1e-4)
    ha = np.genfromtxt(outs["plain"] + "_history.csv", delimiter=",", names=True)
    hb = np.genfromtxt(outs["rccl"] + "_history.csv", delimiter=",", names=True)
    assert np.all(np.isfinite(hb["loss"])) and np.allclose(ha["loss"], hb["loss"], rtol=1e-5)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (skips on the driver's one-GPU box; runs the day a node exists)")
@pytest.mark.parametrize("split", ["rows", "owners"])
def test_two_ranks_over_rccl_reproduce_the_one_rank_history(split):
    """Two ranks on two GPUs over RCCL / xGMI, both splits of the observations (rows: the default; reflection owners: opt-in until this
    has run on a node, engine.py): `ranks_seen == 2` and the loss history of the one-rank run on the same problem (noise keyed by global
    indices: the trajectory does not depend on the GPU count beyond summation order)."""
    argv = ["--nobs", "2000000", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--extra", "none"]
    one = _bench(argv, _env())
    two = _bench(argv + ["--gpus", "2"], _env(CARELESS_HIP_OWNER_SHARD="1" if split == "owners" else "0"), timeout=1800)
    assert two["ranks_seen"] == 2 and two["n_gpus"] == 2 and two["backend"] == "nccl"
    assert ("reflection-owner" in two["config"]["parallelism"]) == (split == "owners")
    assert sum(two["obs_per_rank"]) == 2000000
    assert np.allclose(two["loss_history"], one["loss_history"], rtol=1e-5)
