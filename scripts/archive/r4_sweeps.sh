#!/bin/bash
# round 4: larger seeded random sweeps of the parity tests on the final build (other seeds than the default suite's): evidence, not part of the suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for seed in 404 505 606; do
  ENGINE_RANDOM_N=200 ENGINE_RANDOM_SEED=$seed timeout 1500 python -m pytest tests/test_gpu_parity.py -q --no-header -k random_engine 2>&1 | tail -3 | sed "s/^/engine seed $seed: /"
  LANE_RANDOM_N=80 LANE_RANDOM_SEED=$seed LANE_ROWS_RANDOM_N=80 LANE_ROWS_RANDOM_SEED=$seed timeout 1500 python -m pytest tests/test_gpu_parity.py -q --no-header -k random_shapes 2>&1 | tail -3 | sed "s/^/lane seed $seed: /"
done
} | tee gpurun_out/r4_random_sweeps.txt
