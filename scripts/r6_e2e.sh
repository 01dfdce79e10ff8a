#!/bin/bash
# Round 6: --merge-half-datasets through the command line on 5 M observations (as profiles/r5_e2e_format.txt measured it in round 5):
# 2 000 iterations for the main training and for each half, scaler frozen in the halves -- now on cl_frozen_rows.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
mkdir -p gpurun_out/r6
{
echo "# default scaler, --merge-half-datasets, 5 M observations, 2000 iterations";
E2E_TMP=/tmp/e2e timeout 1500 python3 scripts/e2e_profile.py 5000000 2000 --merge-half-datasets 2>/dev/null | grep -E "TOTAL|train_model|format_files|output_step" | head -6
echo "# ... with round 5's slot kernels (FROZEN_SORTED_ROWS off is a class switch: here the whole short cut off = the fused step for the halves)";
CARELESS_HIP_FROZEN_FAST=0 E2E_TMP=/tmp/e2e timeout 1500 python3 scripts/e2e_profile.py 5000000 2000 --merge-half-datasets 2>/dev/null | grep -E "TOTAL|train_model" | head -3
echo "# 5 x 64, 8 samples, Student-t";
E2E_TMP=/tmp/e2e timeout 1500 python3 scripts/e2e_profile.py 5000000 2000 --merge-half-datasets --mlp-layers 5 --mlp-width 64 --mc-samples 8 --studentt-likelihood-dof 16 2>/dev/null | grep -E "TOTAL|train_model" | head -3
echo "# --mlp-layers 10 (the lane kernel compiled for depth 10), plain run";
E2E_TMP=/tmp/e2e timeout 1500 python3 scripts/e2e_profile.py 5000000 2000 --mlp-layers 10 2>/dev/null | grep -E "TOTAL|train_model" | head -3
} | tee gpurun_out/r6/e2e.txt
