"""`python bench.py --gpus N` without a launcher environment starts the N ranks itself (bench.py: launch).  CPU test of that
control flow with a stand-in rank program: N fresh processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, a gloo rendezvous
among them, rank 0's JSON line relayed, non-zero exit when a rank dies or when the world that came up is not N."""
import json
import os
import subprocess
import sys
import textwrap

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent("""
    import json, os, sys
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode = sys.argv[sys.argv.index("--workload") + 1]
    if mode == "die" and rank == 1:
        sys.exit(3)
    if mode in ("die", "hang"):              # rank 0 would wait (for rank 1 / in a collective) forever: the parent has to end it
        import time; time.sleep(600)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    seen = dist.get_world_size() if mode != "lie" else 1
    dist.barrier(); dist.destroy_process_group()
    print("noise on stdout of rank", rank)
    if rank == 0:
        out = {"n_gpus": world, "ranks_seen": seen, "sum": float(t), "local_rank": os.environ["LOCAL_RANK"]}
        if mode == "extra_skipped":
            out["extra_failed"] = "dw_50M_normal_5x64_S1 skipped (host memory)"
        print(json.dumps(out))
""")


def _launch(tmp_path, mode, n=2, extra_args=()):
    stub = tmp_path / "stub_rank.py"
    stub.write_text(STUB)
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        argv = ["--gpus", "{n}", "--workload", "{mode}"] + {list(extra_args)!r}
        sys.exit(bench.launch(bench.parse(argv), argv, script={str(stub)!r}))
    """)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)


def test_launcher_starts_n_ranks_and_relays_rank0_json(tmp_path):
    r = _launch(tmp_path, "ok", n=3)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                   # ONE JSON line on stdout, nothing else
    out = json.loads(lines[0])
    assert out == {"n_gpus": 3, "ranks_seen": 3, "sum": 6.0, "local_rank": "0"}


def test_launcher_fails_when_a_rank_dies_or_the_world_is_short(tmp_path):
    r = _launch(tmp_path, "die")
    assert r.returncode != 0 and "rank 1 exited with code 3" in r.stderr and not r.stdout.strip()
    r = _launch(tmp_path, "lie")
    assert r.returncode != 0 and "asked for 2 ranks" in r.stderr and not r.stdout.strip()


def test_launcher_times_out_and_reports_an_unmeasured_extra_configuration(tmp_path):
    """Ranks stuck in mismatched collectives must not hold the launcher forever (--launch-timeout); a line whose extra configuration
    (the one BASELINE.json quotes at this GPU count) was skipped or failed is relayed, but the exit code is non-zero."""
    r = _launch(tmp_path, "hang", extra_args=("--launch-timeout", "3"))
    assert r.returncode != 0 and "--launch-timeout" in r.stderr and not r.stdout.strip()
    r = _launch(tmp_path, "extra_skipped")
    assert r.returncode != 0 and "extra configuration not measured" in r.stderr
    assert json.loads(r.stdout.strip())["extra_failed"].startswith("dw_50M")


def test_worker_refuses_a_world_that_is_not_gpus(monkeypatch):
    """`--gpus 8` inside a 1-rank environment (or the reverse) must not print an n_gpus=1 line with rc 0"""
    import pytest
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "1"); monkeypatch.setenv("LOCAL_RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8", "--no-cpu-baseline"])
    assert "WORLD_SIZE=1" in str(e.value)


def test_cpu_baseline_runs_in_a_child_process_and_a_dead_child_only_drops_the_baseline(monkeypatch):
    """`bench.py` times the CPU baseline in a FRESH child (`--cpu-child`, sized to the cgroup's memory): the child prints one JSON object; a
    child that dies (the host killing it for memory) leaves a `cpu_baseline` that says so instead of taking the bench line with it."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-child", json.dumps(["mono_1M_normal_5x64_S1", 5000, 1, False])],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1000:]
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["kind"] == "port" and d["n_obs"] == 5000 and d["value"] > 0 and d["cores"] >= 1
    assert bench._host_free_bytes() > 0

    class Dead:
        returncode, stdout, stderr = -9, "", "Killed"
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: Dead())
    out = bench.cpu_baseline_child("mono_1M_normal_5x64_S1", 10_000_000, 5, True)
    assert out["value"] is None and "not measured" in out["sample"] and "-9" in out["sample"]


def test_scale_curve_script_parses():
    assert subprocess.run(["bash", "-n", os.path.join(ROOT, "scripts", "scale_curve.sh")]).returncode == 0
    for s in ("profiles_all.sh", "pmc_passes_step.sh", "store_profiles.sh"):
        assert subprocess.run(["bash", "-n", os.path.join(ROOT, "scripts", s)]).returncode == 0, s
