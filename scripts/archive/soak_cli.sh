#!/bin/bash
# gpurun helper: the reference's command line on its own PYP fixture for 3000 iterations with the default scaler, once on the
# lane-per-observation kernel and once on elbo_narrow.hip: both must stay finite and end at the same loss to a few 1e-3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak
for v in 1 0; do
  CARELESS_HIP_LANE=$v timeout 900 python -m careless_amd mono --iterations 3000 --disable-progress-bar "dHKL,Hobs,Kobs,Lobs,BATCH" tests/golden/pyp_off.mtz gpurun_out/soak/lane$v > gpurun_out/soak/log$v.txt 2>&1 || tail -5 gpurun_out/soak/log$v.txt
  python - <<PY
import csv, math
rows = list(csv.DictReader(open("gpurun_out/soak/lane$v" + "_history.csv")))
loss = [float(r["loss"]) for r in rows]
print("LANE=$v steps", len(loss), "finite", all(math.isfinite(x) for x in loss), "first %.4f last %.4f min %.4f" % (loss[0], loss[-1], min(loss)))
PY
done
