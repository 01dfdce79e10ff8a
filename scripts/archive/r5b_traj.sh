cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5/soak3
for k in 1 2 3; do python -m careless_amd mono --iterations 2000 --disable-progress-bar --image-layers 2 dHKL,image_id,X,Y tests/golden/pyp_off.mtz gpurun_out/r5/soak3/a$k > /dev/null 2>&1; done
for k in 1 2; do CARELESS_HIP_LANE=0 python -m careless_amd mono --iterations 2000 --disable-progress-bar --image-layers 2 dHKL,image_id,X,Y tests/golden/pyp_off.mtz gpurun_out/r5/soak3/b$k > /dev/null 2>&1; done
for k in 1 2; do python -m careless_amd mono --iterations 2000 --disable-progress-bar dHKL,image_id,X,Y tests/golden/pyp_off.mtz gpurun_out/r5/soak3/c$k > /dev/null 2>&1; done
python - <<'PY'
import numpy as np
def L(n): return np.genfromtxt(f"gpurun_out/r5/soak3/{n}_history.csv", delimiter=",", names=True)["loss"]
for grp in (("a1","a2","a3"),("b1","b2"),("c1","c2")):
    ls=[L(n) for n in grp]
    print(grp, "len", [len(l) for l in ls], "finite", [bool(np.isfinite(l).all()) for l in ls])
    for s in (0,1,5,20,100,300,1000,1999):
        print("   step",s,[("%.6g"%l[s]) if s < len(l) else "-" for l in ls])
PY
