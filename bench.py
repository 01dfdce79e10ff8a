#!/usr/bin/env python3
"""bench.py -- reflections/sec per ELBO step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-cpu-baseline]

One "step" = one full-batch ELBO step (forward + backward + gradient norm + Adam) over all observations of the
workload -- what the reference runs per iteration of `train_model` (careless/models/merging/variational.py:255-256).
Workload at every N: BASELINE.json configs[2] (10 M observations, Student-T likelihood, positional-encoding metadata,
5x64 scaler, mc-samples 8), observations sharded over the ranks (strong scaling), one all-reduce of the flat gradient.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="mono_10M_studentt_posenc_5x64_S8")
    ap.add_argument("--nobs", type=int, default=None, help="override the number of observations (debugging)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the gradient all-reduce even with one rank")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank "
                                                      "control flow on a single-GPU box)")
    ap.add_argument("--cpu-sample", type=int, default=200_000)
    ap.add_argument("--sim-world", type=int, default=0, help="diagnostic: run rank 0's shard of a W-rank job on this one GPU "
                                                             "(per-rank step time of the strong-scaling runs; not a bench line)")
    return ap.parse_args()


def cpu_baseline(workload: str, n_sample: int):
    """The oracle (fp32 PyTorch-CPU restatement of the reference graph -- NOT TensorFlow) timed on this box's host
    cores on a bounded sample of the same workload.  Reported beside the GPU number, never the thing shipped."""
    import torch
    from careless_amd.workloads import WORKLOADS
    from oracle import elbo_oracle as O
    spec = WORKLOADS[workload]
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    data = O.make_synthetic(n_sample, d0=spec["d0"], posenc=spec["posenc"], outliers=spec["outliers"])
    cfg = O.ElboConfig(mc_samples=spec["S"], likelihood="normal" if spec["dof"] is None else "studentt", dof=spec["dof"])
    dt = torch.float32
    x = O.inputs_from_numpy(data, dtype=dt)
    p = O.init_params(data, cfg, spec["L"], spec["w"], dtype=dt)
    st = O.AdamState.zeros_like(p.tensors())
    g = torch.Generator().manual_seed(0)
    R, S = int(data["n_refl"]), spec["S"]
    def one_step():
        u = torch.rand(S, R, generator=g, dtype=dt).clamp(1e-6, 1 - 1e-6)
        eta = torch.randn(S, n_sample, generator=g, dtype=dt)
        t0 = time.perf_counter()
        O.train_step(p, x, cfg, st, u, eta)
        return time.perf_counter() - t0

    # be fair to the CPU: more threads than physical cores (or than the cgroup grants) only slows torch down, so try
    # a few thread counts on one step each and keep the fastest for the timed run
    best, cores = None, avail
    for nt in sorted({avail, min(avail, 64), min(avail, 32), min(avail, 16)}, reverse=True):
        torch.set_num_threads(nt)
        one_step()
        tt = one_step()
        if best is None or tt < best:
            best, cores = tt, nt
    torch.set_num_threads(cores)
    t = float(np.median([one_step() for _ in range(3)]))
    return {"value": n_sample / t, "unit": "reflections/s", "cores": cores, "kind": "port",
            "sample": f"{n_sample} observations of the same workload, median of 3 steps after warm-up, {cores} torch threads (best of a small sweep), "
                      f"fp32 PyTorch-CPU restatement of the reference graph (not TensorFlow), torch {torch.__version__}"}


def traffic_bytes(workload: str, world: int):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic.json; collected with
    scripts/pmc_passes.sh, separate --pmc runs); None for configurations that were not profiled."""
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        rec = json.load(open(os.path.join(here, "profiles", "traffic.json"))).get(workload)
        if rec and rec.get("n_gpus") == world:
            return rec["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from careless_amd.workloads import flops_per_obs, bytes_per_obs, make_workload

    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        # RCCL writes its NCCL_DEBUG chatter (version banner, warnings) to stdout through C stdio, where it interleaves with
        # the JSON line: send it to a file instead
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rccl_bench_%h_%p.log")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    model, inputs, data, spec = make_workload(args.workload, N=args.nobs)
    if args.sim_world > 1:
        model.set_data_parallel(0, args.sim_world)
    elif use_dist:
        model.set_data_parallel(rank, world)
    eng = model.engine(inputs)
    eng.force_allreduce = bool(args.force_dist)
    steps_total = args.warmup + args.steps
    eng.alloc_history(steps_total)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        eng.train_step(i)
    sync()
    # the dominant kernel is timed live with events on the stream it is launched on (torch's current stream)
    # the fused scaler kernel: one launch per step (mono), or forward + backward launches around the harmonic sums (Laue)
    timed_names = ("cl_elbo_mono_fwd_bwd", "cl_mlp_forward", "cl_mlp_backward_ext")
    # (Laue: one launch on the single-pass path, forward + backward launches on the two-pass fallback)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps * 2)]
    slot = {"i": 0}

    def timed(fn):
        def call(*a):
            e0, e1 = ev[slot["i"]]
            e0.record()
            rc = fn(*a)
            e1.record()
            slot["i"] += 1
            return rc
        return call

    class _LibProxy:
        def __init__(self, lib):
            self._lib = lib

        def __getattr__(self, k):
            return timed(getattr(self._lib, k)) if k in timed_names else getattr(self._lib, k)

    real_lib = eng.lib
    eng.lib = _LibProxy(real_lib)
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.train_step(args.warmup + i)
    sync()
    t1 = time.perf_counter()
    eng.lib = real_lib
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    kern_ms = float(np.sum([a.elapsed_time(b) for a, b in ev[:slot["i"]]])) / args.steps        # fused-kernel time per step
    launches_per_step = slot["i"] // args.steps
    hist = eng.read_history(steps_total)
    finite = bool(np.all(np.isfinite(hist["loss"]))) and len(hist["loss"]) == steps_total

    out = None
    if rank == 0:
        N = spec["N"]
        ms = 1e3 * elapsed / args.steps
        F = flops_per_obs(spec["d"], spec["w"], spec["L"], spec.get("image_layers", 0))
        B = bytes_per_obs(spec["d"], spec["S"])
        achieved = F * eng.N / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "reflections/sec per ELBO step", "value": N / (elapsed / args.steps), "unit": "reflections/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "n_obs": N, "n_refl": spec["R"], "n_images": spec["M"],
                       "metadata_width": spec["d"], "mlp": f"{spec['L']}x{spec['w']}", "mc_samples": spec["S"],
                       "likelihood": "normal" if spec["dof"] is None else f"studentt(dof={spec['dof']})",
                       "prior": "double-wilson (2 ASUs, r=0.9)" if spec.get("kind") == "double_wilson" else "wilson",
                       "kind": spec.get("kind", "mono"), "image_scales": spec.get("image_layers", 0) == 0,
                       "image_layers": spec.get("image_layers", 0), "noise": "in-kernel philox",
                       "parallelism": f"obs-shard x{world}" if world > 1 else "single",
                       "loss_finite": finite, "final_loss": hist["loss"][-1] if hist["loss"] else None},
            "roofline": {"bound": "mfma", "kernel": "elbo_mlp_kernel (" + ("cl_mlp_forward + cl_mlp_backward_ext" if launches_per_step == 2 else "cl_elbo_mono_fwd_bwd") + ")", "achieved": achieved,
                         "peak": 157.3, "unit": "TFLOP/s", "frac": achieved / 157.3, "traffic": traffic_bytes(args.workload, world),
                         "kernel_ms": kern_ms, "flops_per_obs": F, "obs_per_launch": eng.N,
                         "hbm_secondary": {"achieved_GBps": B * eng.N / (kern_ms * 1e-3) / 1e9, "bytes_per_obs": B}},
        }
        if args.sim_world > 1:
            out["diagnostic"] = f"rank 0 shard of a simulated {args.sim_world}-rank job: value is NOT a throughput of this workload"
            out["value"] = None
        if not args.no_cpu_baseline and world == 1 and spec.get("kind", "mono") == "mono" and not spec.get("image_layers"):
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        try:                                        # push out whatever RCCL left in C stdio (version banner) first
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)          # the ONE JSON line, last on stdout


if __name__ == "__main__":
    main()
