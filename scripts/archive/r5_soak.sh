#!/bin/bash
# gpurun helper (round 5): soak of the new paths through the command line on the reference's MTZ fixture -- 3 000 iterations each:
# four positionally encoded keys (peeled first layer), hidden width 16, and the default for comparison; loss must fall and stay finite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5/soak
F=tests/golden/pyp_off.mtz
run() {
  name=$1; shift
  timeout 900 python -m careless_amd mono --iterations 3000 --disable-progress-bar "$@" $F gpurun_out/r5/soak/$name > gpurun_out/r5/soak/$name.log 2>&1
  python - gpurun_out/r5/soak/${name}_history.csv $name <<'PY'
import sys, numpy as np
h = np.genfromtxt(sys.argv[1], delimiter=",", names=True)
l = h["loss"]
print("%-28s steps %d finite %s loss first %.4g @300 %.4g @1000 %.4g last %.4g  min grad norm %.3g" % (sys.argv[2], len(l), bool(np.all(np.isfinite(l))), l[0], l[300], l[1000], l[-1], h["Grad_Norm"].min() if "Grad_Norm" in h.dtype.names else float("nan")))
PY
}
run default dHKL,image_id,X,Y
run posenc4 --positional-encoding-keys X,Y,Hobs,Kobs --mc-samples 4 --studentt-likelihood-dof 16 dHKL,image_id,X,Y,Hobs,Kobs
run width16 --mlp-width 16 --mlp-layers 10 dHKL,image_id,X,Y
run width12_d38 --mlp-width 12 --mlp-layers 8 --positional-encoding-keys X,Y,Hobs,Kobs dHKL,image_id,X,Y,Hobs,Kobs
