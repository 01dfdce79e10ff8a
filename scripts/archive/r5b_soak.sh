#!/bin/bash
# gpurun helper (round 5, second half): soak of the new paths through the command line on the reference's MTZ fixtures -- 3 000 iterations each:
# --image-layers on the default scaler (lane kernel with per-image layers), the same with --merge-half-datasets (frozen-scaler steps),
# --freeze-scales from saved weights, Laue data with image layers; loss must fall and stay finite, the half-dataset merges must correlate
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5/soak2
report() {
  python - gpurun_out/r5/soak2/${1}_history.csv $1 <<'PY'
import sys, numpy as np
h = np.genfromtxt(sys.argv[1], delimiter=",", names=True)
l = h["loss"]
print("%-28s steps %d finite %s loss first %.4g @300 %.4g @1000 %.4g last %.4g" % (sys.argv[2], len(l), bool(np.all(np.isfinite(l))), l[0], l[min(300, len(l) - 1)], l[min(1000, len(l) - 1)], l[-1]))
PY
}
run() { name=$1; mode=$2; shift; shift; timeout 1200 python -m careless_amd $mode --iterations 3000 --disable-progress-bar "$@" gpurun_out/r5/soak2/$name > gpurun_out/r5/soak2/$name.log 2>&1 || tail -3 gpurun_out/r5/soak2/$name.log; report $name; }
run img2 mono --image-layers 2 dHKL,image_id,X,Y tests/golden/pyp_off.mtz
run img2_halves mono --image-layers 2 --merge-half-datasets --half-dataset-repeats 2 dHKL,image_id,X,Y tests/golden/pyp_off.mtz
run halves mono --merge-half-datasets --studentt-likelihood-dof 16 --mc-samples 4 dHKL,image_id,X,Y tests/golden/pyp_off.mtz
run frozen mono --scale-file gpurun_out/r5/soak2/halves_scale --freeze-scales dHKL,image_id,X,Y tests/golden/pyp_off.mtz
run laue_img1 poly --image-layers 1 --merge-half-datasets dHKL,image_id,X,Y,Wavelength tests/golden/pyp_2ms.mtz
python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from careless_amd.io.mtz import read_mtz
for name in ("img2_halves", "halves", "laue_img1"):
    m = read_mtz(f"gpurun_out/r5/soak2/{name}_xval_0.mtz")
    c = m.columns
    key = c["H"].astype(np.int64) * 1000003 + c["K"].astype(np.int64) * 1009 + c["L"].astype(np.int64)
    rep0 = c["repeat"] == 0
    a = {k: f for k, f, h, r in zip(key, c["F"], c["half"], rep0) if h == 0 and r}
    b = {k: f for k, f, h, r in zip(key, c["F"], c["half"], rep0) if h == 1 and r}
    ks = sorted(set(a) & set(b))
    cc = np.corrcoef([a[k] for k in ks], [b[k] for k in ks])[0, 1]
    print("%-28s xval rows %d, reflections in both halves %d, CC(F half 1, F half 2) = %.4f" % (name, len(m), len(ks), cc))
PY
