import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from careless_amd.careless import run_careless
from careless_amd.parser import parser
import tempfile
td = tempfile.mkdtemp()
args = parser.parse_args(f"mono --iterations=3000 --disable-progress-bar --studentt-likelihood-dof=16 --mc-samples=4 dHKL,image_id,Hobs,Kobs,Lobs tests/golden/pyp_off.mtz {td}/out".split())
model, hist = run_careless(args)
l = np.array(hist["loss"]); g = np.array(hist["Grad Norm"])
print("steps", len(l), "loss first/last", l[0], l[-1], "min", l.min(), "finite", np.isfinite(l).all(), "gradnorm last", g[-1])
from careless_amd.io.mtz import read_mtz
m = read_mtz(f"{td}/out_0.mtz"); print("merged reflections", len(m), "mean F/SigF", float(np.mean(m.columns["F"]/m.columns["SigF"])))
