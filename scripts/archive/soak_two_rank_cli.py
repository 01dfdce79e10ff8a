#!/usr/bin/env python3
"""300 iterations of `careless_amd mono` on the reference's MTZ fixture as ONE process and as TWO gloo ranks (rows, then reflection
owners) sharing this GPU: merged amplitudes and histories must agree (in-kernel noise is keyed by global indices, so the trajectory
does not depend on the split; what differs is the summation order of float atomics)."""
import os, socket, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from careless_amd.io.mtz import read_mtz
PYP = os.path.join(ROOT, "tests", "golden", "pyp_off.mtz")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
flags = f"mono --iterations={N} --disable-progress-bar --mlp-layers 3 --test-fraction 0.2 dHKL,image_id".split()
tmp = tempfile.mkdtemp()
env0 = dict(os.environ, PYTHONPATH=ROOT)
one = os.path.join(tmp, "one")
subprocess.check_call([sys.executable, "-m", "careless_amd"] + flags + [PYP, one], env=env0, cwd=ROOT, stdout=subprocess.DEVNULL)
a = read_mtz(one + "_0.mtz")
ha = np.genfromtxt(one + "_history.csv", delimiter=",", names=True)
for split in ("rows", "owners"):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = os.path.join(tmp, split)
    procs = []
    for r in range(2):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CARELESS_DIST_BACKEND="gloo", CARELESS_HIP_OWNER_SHARD="1" if split == "owners" else "0")
        procs.append(subprocess.Popen([sys.executable, "-m", "careless_amd"] + flags + [PYP, out], env=env, cwd=ROOT, stdout=subprocess.DEVNULL))
    assert [p.wait(timeout=900) for p in procs] == [0, 0]
    b = read_mtz(out + "_0.mtz")
    hb = np.genfromtxt(out + "_history.csv", delimiter=",", names=True)
    dF = np.max(np.abs(a.columns["F"] - b.columns["F"]) / np.abs(a.columns["F"]))
    dS = np.max(np.abs(a.columns["SigF"] - b.columns["SigF"]) / np.abs(a.columns["SigF"]))
    dl = np.max(np.abs(ha["loss"] - hb["loss"]) / np.abs(ha["loss"]))
    dv = np.nanmax(np.abs(ha["NLL_val"] - hb["NLL_val"]) / np.abs(ha["NLL_val"]))
    print(f"{split:7s}: {N} iterations, 2 ranks vs 1: max rel diff F {dF:.2e} SigF {dS:.2e} loss history {dl:.2e} NLL_val {dv:.2e}; final loss {hb['loss'][-1]:.6f} vs {ha['loss'][-1]:.6f}")
    assert dF < 1e-3 and dl < 1e-4, split
print("ok")
