#!/usr/bin/env python3
"""bench.py -- reflections/sec per ELBO step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-cpu-baseline]

One "step" = one full-batch ELBO step (forward + backward + gradient norm + Adam) over all observations of the
workload -- what the reference runs per iteration of `train_model` (careless/models/merging/variational.py:255-256).
Workload at every N: BASELINE.json configs[2] (10 M observations, Student-T likelihood, positional-encoding metadata,
5x64 scaler, mc-samples 8), observations sharded over the ranks (strong scaling), one all-reduce of the flat gradient.
Prints ONE JSON line (rank 0).

Ranks: one process per GPU.  Under `python -m torch.distributed.run ... bench.py --gpus N` the ranks exist already
(RANK / LOCAL_RANK / WORLD_SIZE in the environment).  A bare `python bench.py --gpus N` starts them itself: the parent
process -- which never touches the GPU, before or after -- spawns N fresh children with that environment, relays rank 0's
JSON line and exits non-zero if any rank fails or if the world that came up is not N.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

HEADLINE = "mono_10M_studentt_posenc_5x64_S8"
# BASELINE.json quotes configs[3] on 4 GPUs and configs[4] on 8: measured after the headline line when that many ranks are up
EXTRA_AT = {4: "laue_5M_normal_5x64_S1", 8: "dw_50M_normal_5x64_S1"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=HEADLINE)
    ap.add_argument("--nobs", type=int, default=None, help="override the number of observations (debugging)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the gradient all-reduce even with one rank")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank "
                                                      "control flow on a single-GPU box)")
    ap.add_argument("--cpu-sample", type=int, default=None, help="time the CPU baseline on a bounded sample of this many observations "
                                                                 "(3 steps) instead of the workload's full size")
    ap.add_argument("--cpu-steps", type=int, default=None)
    ap.add_argument("--cpu-full", action="store_true", help="(the default since round 4) CPU baseline at the workload's full size, 5 timed "
                                                            "steps, as SURVEY 8d prescribes: ~2 minutes and ~50 GB of host memory for the "
                                                            "headline workload; cut to what MemAvailable holds, and said so, on a smaller host")
    ap.add_argument("--extra", default="auto", help="'auto': with 4 (8) ranks also time the Laue (double-Wilson) configuration "
                                                    "BASELINE.json quotes on 4 (8) GPUs and report it under 'extra_configs'; 'none'; or a workload name")
    ap.add_argument("--extra-nobs", type=int, default=None, help="observations of the extra configuration (rehearsals on a small box)")
    ap.add_argument("--launch-timeout", type=float, default=3600.0, help="seconds after which the launcher (bare `bench.py --gpus N`) "
                                                                         "terminates its ranks and exits non-zero")
    ap.add_argument("--sim-world", type=int, default=0, help="diagnostic: run rank 0's shard of a W-rank job on this one GPU "
                                                             "(per-rank step time of the strong-scaling runs; not a bench line)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------------------
# launcher: parent of the N rank processes (no torch.cuda / HIP call in this process, ever)
# ----------------------------------------------------------------------------------------------------------------------------
def launch(args, argv, script=None) -> int:
    """`script`: the program every rank runs (this file; the launcher's unit test passes a CPU stand-in)."""
    import socket
    import subprocess
    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        # rank 0's stdout carries the JSON line; the other ranks' stdout goes to our stderr so nothing can interleave with it
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    failed = None
    t_start = time.monotonic()
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
        if failed is None and time.monotonic() - t_start > args.launch_timeout:
            failed = (-1, "timeout")             # ranks stuck (e.g. in mismatched collectives): never wait forever
        time.sleep(0.2)
    if failed is None:
        for r, p in enumerate(procs):
            if p.returncode != 0:
                failed = (r, p.returncode)
    if failed is not None:                       # the others may be waiting for the dead rank inside a collective
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:
                p.kill()
        print(f"bench.py: rank {failed[0]} exited with code {failed[1]}" if failed[0] >= 0 else
              f"bench.py: ranks still running after --launch-timeout {args.launch_timeout:.0f} s: terminated", file=sys.stderr)
        return 1
    reader.join(timeout=30)
    out = None
    for ln in lines:
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                out = json.loads(ln)
            except ValueError:
                pass
    if out is None:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        return 1
    if out.get("n_gpus") != n or out.get("ranks_seen") != n:
        print(f"bench.py: asked for {n} ranks, the job saw {out.get('ranks_seen')}", file=sys.stderr)
        return 1
    print(json.dumps(out), flush=True)
    if out.get("extra_failed"):                  # the configuration BASELINE.json quotes at this GPU count did not get measured
        print(f"bench.py: extra configuration not measured: {out['extra_failed']}", file=sys.stderr)
        return 1
    return 0


# ----------------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ----------------------------------------------------------------------------------------------------------------------------
def cpu_baseline(workload: str, n_sample: int, steps: int, full: bool = False):
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
    dt = torch.float32
    cfg = O.ElboConfig(mc_samples=spec["S"], likelihood="normal" if spec["dof"] is None else "studentt", dof=spec["dof"])
    S = spec["S"]

    def problem(n):
        data = O.make_synthetic(n, d0=spec["d0"], posenc=spec["posenc"], outliers=spec["outliers"], posenc_keys=spec.get("posenc_keys", 2))
        x = O.inputs_from_numpy(data, dtype=dt)
        p = O.init_params(data, cfg, spec["L"], spec["w"], dtype=dt)
        st = O.AdamState.zeros_like(p.tensors())
        g = torch.Generator().manual_seed(0)
        R = int(data["n_refl"])

        def one_step():
            u = torch.rand(S, R, generator=g, dtype=dt).clamp(1e-6, 1 - 1e-6)
            eta = torch.randn(S, n, generator=g, dtype=dt)
            t0 = time.perf_counter()
            O.train_step(p, x, cfg, st, u, eta)
            return time.perf_counter() - t0
        return one_step

    # thread count: more torch threads than physical cores (or than the cgroup grants) only slows the CPU path down.  Picked
    # on a small separate problem, outside the timed run
    probe = problem(100_000)
    best, cores = None, avail
    for nt in sorted({avail, min(avail, 64), min(avail, 32), min(avail, 16)}, reverse=True):
        torch.set_num_threads(nt)
        probe()
        tt = probe()
        if best is None or tt < best:
            best, cores = tt, nt
    del probe
    torch.set_num_threads(cores)
    n_want = n_sample
    if full:
        steps = max(5, steps)
        # autograd keeps ~(2 L w + 14 S) fp32 values per observation alive; stay inside the host's free memory
        per_obs = 4 * (3 * spec["L"] * spec["w"] + 24 * S)
        n_sample = int(min(n_sample, 0.6 * _host_free_bytes() / per_obs))
    step = problem(n_sample)
    step()                                            # warm-up (allocator, thread pool)
    times = [step() for _ in range(steps)]
    t = float(np.median(times))
    size = "full size" if n_sample == spec["N"] else ("bounded sample" if n_sample == n_want else
                                                      f"bounded sample: host memory holds {n_sample} of the {n_want} asked for")
    return {"value": n_sample / t, "unit": "reflections/s", "cores": cores, "kind": "port", "seconds_per_step": t, "n_obs": n_sample,
            "sample": f"{n_sample} observations of the same workload ({size}), median of "
                      f"{steps} steps after one warm-up step, {cores} torch threads (picked on a separate 100k problem), "
                      f"fp32 PyTorch-CPU restatement of the reference graph (not TensorFlow), torch {torch.__version__}"}


def _host_free_bytes() -> float:
    """Memory this process may still take: MemAvailable, cut to what the cgroup (container / lease) leaves -- /proc/meminfo does not see
    that limit, and a baseline sized past it is killed without a word."""
    free = float("inf")
    try:
        free = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024.0
    except Exception:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            v = open(lim).read().strip()
            if v != "max" and int(v) < (1 << 60):
                free = min(free, float(int(v) - int(open(cur).read().strip())))
        except Exception:
            pass
    return free


def cpu_baseline_child(workload: str, n_sample: int, steps: int, full: bool):
    """The CPU baseline in a FRESH child process (no GPU in it): if the host kills it -- memory -- or it fails, the bench line keeps its
    GPU result and says what happened, and a bounded 2 M-observation sample is tried instead."""
    import subprocess
    here = os.path.abspath(__file__)
    for n, st, fl in ((n_sample, steps, full), (min(n_sample, 2_000_000), 3, False)):
        try:
            r = subprocess.run([sys.executable, here, "--cpu-child", json.dumps([workload, n, st, fl])], capture_output=True, text=True, timeout=3600)
            if r.returncode == 0:
                return json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
            err = f"exit code {r.returncode}: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ''}"
        except Exception as e:       # noqa: BLE001
            err = repr(e)
        print(f"bench.py: CPU baseline on {n} observations failed ({err})", file=sys.stderr, flush=True)
        if n <= 2_000_000:
            break
    return {"value": None, "unit": "reflections/s", "cores": None, "kind": "port", "sample": f"not measured: {err}"}


def _traffic_from(workload: str, world: int):
    """Where `roofline.traffic` comes from: the source hash of the kernels the PMC passes ran on and whether it is this build's."""
    from careless_amd.build import source_hash
    rec = traffic_bytes(workload, world)[1]
    if rec is None:
        return None
    return {"sources": rec.get("sources"), "this_build": source_hash(), "stale": rec.get("sources") != source_hash(), "file": rec.get("file")}


def traffic_bytes(workload: str, world: int):
    """(HBM bytes per launch of the dominant kernel, record) from the committed PMC passes (profiles/traffic.json; collected with
    scripts/pmc_passes.sh, separate --pmc runs; written by scripts/traffic_json.py together with the source hash of the kernels it
    was measured on); (None, None) for configurations that were not profiled."""
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        rec = json.load(open(os.path.join(here, "profiles", "traffic.json"))).get(workload)
        if rec and rec.get("n_gpus") == world:
            return rec["hbm_bytes_per_launch"], rec
    except Exception:
        pass
    return None, None


# ----------------------------------------------------------------------------------------------------------------------------
# one timed workload on the ranks that are up
# ----------------------------------------------------------------------------------------------------------------------------
def run_workload(args, name, nobs, steps, warmup, rank, world, use_dist):
    import torch
    import torch.distributed as dist
    from careless_amd.workloads import bytes_per_obs, flops_per_obs, make_workload

    class BuildFailed(RuntimeError):
        pass

    def agree(ok: bool, what: str):
        """All ranks learn whether every rank got through a phase that has no collective of its own (host generation, engine build,
        upload): a rank that failed there must not leave the others waiting in the next collective."""
        if use_dist:
            t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = bool(t.item() > 0.5)
        if not ok:
            raise BuildFailed(what)

    share_dir = None
    if use_dist and world > 1:
        # ONE copy of the synthetic problem per node: rank 0 generates it into /dev/shm, every rank maps it and uploads its own rows
        share_dir = f"/dev/shm/careless_bench_{os.environ.get('MASTER_PORT', '0')}_{name}"
    err = None
    try:
        if share_dir is not None and rank == 0:
            import shutil
            shutil.rmtree(share_dir, ignore_errors=True)       # (left-overs of a killed run)
        gen_ok = {"ok": True}

        def gen_barrier():          # rank 0 is back from the generator (or failed in it): everybody learns which
            agree(gen_ok["ok"], f"{name}: rank 0 could not generate the problem")

        try:
            model, inputs, data, spec = make_workload(name, N=nobs, rank=rank, world=world, share_dir=share_dir,
                                                      barrier=gen_barrier if share_dir else None)
        except BuildFailed:
            raise
        except Exception as e:       # noqa: BLE001  (only rank 0 can fail before the barrier: it generates)
            if share_dir is None or rank != 0:
                raise
            gen_ok["ok"] = False
            err = e
            gen_barrier()
        if args.sim_world > 1:
            model.set_data_parallel(0, args.sim_world)
        elif use_dist:
            model.set_data_parallel(rank, world)
        eng = model.engine(inputs)
    except BuildFailed:
        raise
    except Exception as e:           # noqa: BLE001
        err = e
    agree(err is None, f"{name}: engine build failed on a rank" + (f" (this rank: {err!r})" if err is not None else ""))
    if share_dir is not None:                       # every rank has uploaded its shard: the files can go (the maps die with `data`)
        dist.barrier()
        if rank == 0:
            import shutil
            shutil.rmtree(share_dir, ignore_errors=True)
    del inputs, data
    eng.force_allreduce = bool(args.force_dist)
    steps_total = warmup + steps
    eng.alloc_history(steps_total)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # launches of the dominant kernel(s) per step are counted on the last warm-up step: the event pairs of the timed region are created
    # BEFORE it (creating a few hundred events inside the loop stalled the host once for ~30 ms on long runs)
    n_per_step = {"n": 0}
    # the dominant kernel is timed live with events on the stream it is launched on (torch's current stream):
    # the fused scaler kernel, one launch per step (mono, single-pass Laue), or forward + backward launches around the
    # harmonic sums (two-pass Laue fallback)
    timed_names = ("cl_elbo_mono_fwd_bwd", "cl_mlp_forward", "cl_mlp_backward_ext", "cl_peel_forward", "cl_peel_backward")      # (peeled first layer: its two calls count)
    if eng.wide:                    # width > 64: the layer-by-layer GEMM launches of csrc/wide_gemm.hip are the dominant kernels
        timed_names = ("cl_wide_dense_forward", "cl_wide_dense_forward_head", "cl_wide_dense_forward_head_lik", "cl_wide_dense2_forward", "cl_wide_dense_dgrad", "cl_wide_dense_dgrad_pre",
                       "cl_wide_dense_dgrad_pre_wgrad0", "cl_wide_dense_dgrad_head", "cl_wide_dense_wgrad", "cl_wide_dense_wgrad_pre", "cl_wide_dense_wgrad_head",
                       "cl_wide_head_forward", "cl_wide_head_backward")
    ev = []
    slot = {"i": 0}

    def counted(fn):
        def call(*a):
            n_per_step["n"] += 1
            return fn(*a)
        return call

    def timed(fn):
        def call(*a):
            if slot["i"] >= len(ev):
                ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            e0, e1 = ev[slot["i"]]
            e0.record()
            rc = fn(*a)
            e1.record()
            slot["i"] += 1
            return rc
        return call

    class _LibProxy:
        def __init__(self, lib, wrap):
            self._lib, self._wrap = lib, wrap

        def __getattr__(self, k):
            return self._wrap(getattr(self._lib, k)) if k in timed_names else getattr(self._lib, k)

    real_lib = eng.lib
    for i in range(warmup):
        if i == warmup - 1:
            eng.lib = _LibProxy(real_lib, counted)
        eng.train_step(i)
    ev.extend((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps * n_per_step["n"]))
    sync()
    eng.lib = _LibProxy(real_lib, timed)
    # Python's cyclic garbage collector stays out of the timed region: a generation-2 collection of this process (torch + the
    # problem's host arrays) takes 35 - 55 ms -- more than the 20 timed steps of an 8-rank run together -- and landed inside the timed
    # region of some short runs (scripts/archive/diag_rowsplit.sh: 40 steps of a simulated 8-rank shard, 0.35 -> 1.26 ms per step)
    import gc
    gc.collect()
    gc_was = gc.isenabled()
    if os.environ.get("BENCH_KEEP_GC", "0") != "1":
        gc.disable()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.train_step(warmup + i)
    sync()
    t1 = time.perf_counter()
    if gc_was:
        gc.enable()
    eng.lib = real_lib
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    import resource
    per_rank = torch.tensor([float(eng.N), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2.0 ** 20], dtype=torch.float64, device="cuda")
    obs_per_rank, rss_per_rank = [int(eng.N)], [round(float(per_rank[1].item()), 2)]
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        gathered = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(gathered, per_rank)
        obs_per_rank = [int(g[0].item()) for g in gathered]
        rss_per_rank = [round(float(g[1].item()), 2) for g in gathered]
    elapsed = float(elapsed.item())
    kern_ms = float(np.sum([a.elapsed_time(b) for a, b in ev[:slot["i"]]])) / steps        # fused-kernel time per step
    launches_per_step = slot["i"] // steps
    hist = eng.read_history(steps_total)
    finite = bool(np.all(np.isfinite(hist["loss"]))) and len(hist["loss"]) == steps_total
    if rank != 0:
        return None
    N = spec["N"]
    ms = 1e3 * elapsed / steps
    F = flops_per_obs(spec["d"], spec["w"], spec["L"], spec.get("image_layers", 0))
    B = bytes_per_obs(spec["d"], spec["S"])
    # the label of the dominant kernel comes from the library's own routing (cl_mlp_kernel_name), not from a restatement of it
    kernel_name = eng.kernel_name()
    achieved = F * eng.N / (kern_ms * 1e-3) / 1e12
    achieved_step = F * eng.N / (ms * 1e-3) / 1e12         # SURVEY 8d defines `achieved` on the whole step time
    return {
        "value": N / (elapsed / steps), "ms_per_step": ms, "spec": spec,
        "config": {"workload": name, "n_obs": N, "n_refl": spec["R"], "n_images": spec["M"],
                   "metadata_width": spec["d"], "mlp": f"{spec['L']}x{spec['w']}", "mc_samples": spec["S"],
                   "likelihood": "normal" if spec["dof"] is None else f"studentt(dof={spec['dof']})",
                   "prior": "double-wilson (2 ASUs, r=0.9)" if spec.get("kind") == "double_wilson" else "wilson",
                   "kind": spec.get("kind", "mono"), "image_scales": spec.get("image_layers", 0) == 0,
                   "image_layers": spec.get("image_layers", 0), "noise": "in-kernel philox",
                   "parallelism": (("reflection-owner shard" if eng.owner else "obs-shard") + f" x{eng.shard.world}") if eng.shard.world > 1 else "single",
                   "loss_finite": finite, "final_loss": hist["loss"][-1] if hist["loss"] else None},
        "loss_history": [float(v) for v in hist["loss"]],
        "roofline": {"bound": "mfma", "kernel": (kernel_name + " (" + ("cl_wide_* GEMM launches" if eng.wide else ("cl_peel_forward + cl_elbo_mono_fwd_bwd + cl_peel_backward" if getattr(eng, "peel", False) else ("cl_mlp_forward + cl_mlp_backward_ext" if launches_per_step == 2 else "cl_elbo_mono_fwd_bwd"))) + ")"),
                     "achieved": achieved, "peak": 157.3, "unit": "TFLOP/s", "frac": achieved / 157.3,
                     "traffic": traffic_bytes(name, world)[0], "traffic_from": _traffic_from(name, world), "kernel_ms": kern_ms, "flops_per_obs": F, "obs_per_launch": eng.N,
                     "achieved_on_step_time": achieved_step, "frac_on_step_time": achieved_step / 157.3,
                     "hbm_secondary": {"achieved_GBps": B * eng.N / (kern_ms * 1e-3) / 1e9, "bytes_per_obs": B}},
        "obs_per_rank": obs_per_rank, "host_peak_rss_gib_per_rank": rss_per_rank,
    }


def _build_id():
    from careless_amd.build import source_hash
    return source_hash()


def worker(args) -> int:
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    use_dist = world > 1 or args.force_dist
    if world > 1 and args.backend == "nccl" and ndev < world:
        raise SystemExit(f"--gpus {world} needs {world} visible GPUs, found {ndev} (--backend gloo rehearses the control flow on fewer)")
    torch.cuda.set_device(local_rank % max(1, ndev))
    if use_dist:
        # RCCL writes its NCCL_DEBUG chatter (version banner, warnings) to stdout through C stdio, where it would interleave
        # with the JSON line: send it to a file instead
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rccl_bench_%h_%p.log")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    ranks_seen = dist.get_world_size() if use_dist else 1

    res = run_workload(args, args.workload, args.nobs, args.steps, args.warmup, rank, world, use_dist)
    extra_name = EXTRA_AT.get(world) if args.extra == "auto" else (None if args.extra == "none" else args.extra)
    if args.workload != HEADLINE or args.nobs is not None or args.sim_world > 1:
        extra_name = extra_name if args.extra not in ("auto", "none") else None
    extras, extra_failed = {}, None
    if rank == 0:
        # backup of the headline numbers before anything else runs (stderr: stdout carries exactly ONE JSON line, at the end)
        print("bench.py headline (backup): " + json.dumps({k: res[k] for k in ("value", "ms_per_step", "obs_per_rank")}), file=sys.stderr, flush=True)
    if extra_name:
        # ONE generator peak (~250 B per observation, rank 0) plus the shared copy and the ranks' shard conversions: every rank takes
        # the same decision
        from careless_amd.workloads import WORKLOADS
        n_extra = args.extra_nobs or WORKLOADS[extra_name]["N"]
        need = 400.0 * n_extra
        try:
            avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024.0
        except Exception:
            avail = float("inf")
        if use_dist:
            t = torch.tensor([avail], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            avail = float(t.item())
        if avail < need:
            extras[extra_name] = {"skipped": f"host memory: {avail / 2**30:.0f} GiB available, {need / 2**30:.0f} GiB wanted"}
            extra_failed = f"{extra_name} skipped ({extras[extra_name]['skipped']})"
            extra_name = None
    if extra_name:
        try:                                    # never lose the headline line to the extra configuration
            ex = run_workload(args, extra_name, args.extra_nobs, min(args.steps, 10), min(args.warmup, 2), rank, world, use_dist)
            if ex is not None:
                extras[extra_name] = {"value": ex["value"], "unit": "reflections/s", "n_gpus": world, "ms_per_step": ex["ms_per_step"],
                                      "config": ex["config"], "roofline": ex["roofline"], "obs_per_rank": ex["obs_per_rank"],
                                      "loss_history": ex["loss_history"],
                                      "host_peak_rss_gib_per_rank": ex["host_peak_rss_gib_per_rank"]}
        except Exception as e:                   # noqa: BLE001  (BuildFailed is raised on every rank alike: nobody waits in a collective)
            extras[extra_name] = {"error": repr(e)}
            extra_failed = f"{extra_name} failed: {e!r}"

    out = None
    if rank == 0:
        spec = res.pop("spec")
        out = {"metric": "reflections/sec per ELBO step", "value": res["value"], "unit": "reflections/s",
               "n_gpus": world, "ranks_seen": ranks_seen, "backend": (args.backend if use_dist else None),
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
               "timing": f"mean of {args.steps} steps in one region bracketed by barrier + synchronize (the driver's contract; SURVEY 8d's median of "
                         "single steps would need a host sync per step -- step-to-step jitter is 0.2 %)",
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "build": _build_id(), "config": res["config"], "roofline": res["roofline"], "loss_history": res["loss_history"],
               "obs_per_rank": res["obs_per_rank"],
               "host_peak_rss_gib_per_rank": res["host_peak_rss_gib_per_rank"]}
        if extras:
            out["extra_configs"] = extras
        if extra_failed:
            out["extra_failed"] = extra_failed
        if args.sim_world > 1:
            out["diagnostic"] = f"rank 0 shard of a simulated {args.sim_world}-rank job: value is NOT a throughput of this workload"
            out["value"] = None
        if not args.no_cpu_baseline and world == 1 and spec.get("kind", "mono") == "mono" and not spec.get("image_layers"):
            # full size (configs[2] for >= 5 steps) unless a bounded sample was asked for; cpu_baseline() cuts the size to
            # the host's free memory and says so in `sample`
            full = args.cpu_sample is None
            n_cpu = (args.nobs or spec["N"]) if full else (args.cpu_sample if args.nobs is None else min(args.cpu_sample, args.nobs))
            out["cpu_baseline"] = cpu_baseline_child(args.workload, n_cpu, args.cpu_steps or (5 if full else 3), full)
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
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    if len(argv) == 2 and argv[0] == "--cpu-child":          # the CPU baseline's own process (cpu_baseline_child)
        print(json.dumps(cpu_baseline(*json.loads(argv[1]))), flush=True)
        return 0
    args = parse(argv)
    have_ranks = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not have_ranks:
        return launch(args, argv)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
