#!/bin/bash
# Round 6, after the closing evidence: the regression cases of the withdrawn <16, 64, 24, image layers> instance, then the randomized parity
# sweep of scripts/r6_final_b.sh on three more seeds with larger draws (seed 7 is the one that found the fault).
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
mkdir -p gpurun_out/r6
{
echo "Sources $(python3 -c 'from careless_amd.build import source_hash; print(source_hash())') (scripts/r6_more_sweeps.sh):"
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_routing.py -m gpu -q --no-header -k "d36 or d50 or per_image_layers_run" 2>&1 | tail -2 | tr '\n' ' '; echo " (regression cases of the withdrawn instance + image-layer routing)"
for seed in 7 13 101; do
  ENGINE_RANDOM_SEED=$seed ENGINE_RANDOM_N=100 LANE_DEPTH_RANDOM_N=150 LANE_IMGL_RANDOM_N=60 LANE_IMGL_DEPTH_RANDOM_N=120 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -k random_engine 2>&1 | tail -2 | tr '\n' ' '
  echo " (seed $seed: 100 + 150 + 60 + 120 draws)"
done
} | tee gpurun_out/r6/more_sweeps.txt
