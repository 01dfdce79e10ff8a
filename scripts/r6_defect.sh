#!/bin/bash
# Round 6, first GPU call: the run-to-run defect of the dZ_0-storing lane instances (NOTEBOOK R6.1).
#   1. the hazard itself, issued by hand (scripts/probe/valu_mfma_hazard_probe.hip)
#   2. the withdrawn instance and its diagnostic variants (scripts/probe/build_lane_variants.py), twelve fresh engines each
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
mkdir -p gpurun_out
timeout 300 ./scripts/probe/valu_mfma_hazard_probe > gpurun_out/r6_hazard_probe.txt 2>&1
echo "hazard probe rc=$?" >> gpurun_out/r6_hazard_probe.txt
: > gpurun_out/r6_defect_variants.jsonl
for v in dxo_ni dxo_ni_lrelu_c dxo_ni_sel_c dxo_ni_pad dxo_ni_nostore; do
  lib=careless_amd/lib/variants/libcareless_hip_$v.so
  [ -f "$lib" ] || continue
  CARELESS_HIP_LIB=$PWD/$lib timeout 600 python3 scripts/probe/lane_defect_probe.py --config image_layers2_peeled_d21 --runs 12 --N 5000 --tag $v >> gpurun_out/r6_defect_variants.jsonl 2>gpurun_out/r6_defect_$v.err
  CARELESS_HIP_LIB=$PWD/$lib timeout 600 python3 scripts/probe/lane_defect_probe.py --config image_layers2_peeled_d21 --runs 6 --N 1000000 --images 997 --tag ${v}_1M >> gpurun_out/r6_defect_variants.jsonl 2>>gpurun_out/r6_defect_$v.err
done
tail -c 600 gpurun_out/r6_hazard_probe.txt
python3 - <<'PY'
import json
for ln in open("gpurun_out/r6_defect_variants.jsonl"):
    r = json.loads(ln)
    print(r["tag"], r["kernel"], "bad runs", r["n_bad_runs"], "of", r["runs"] - 1, "distinct nll", r["distinct_nll"],
          "max dz0 rows", max([p.get("dz0_bad_rows", 0) for p in r["per_run"]] or [0]))
PY
