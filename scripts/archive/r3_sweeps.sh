#!/bin/bash
# larger seeded random sweeps of the parity tests (other seeds than the default suite's): evidence, not part of the suite
mkdir -p gpurun_out
{
for seed in 101 202; do
  ENGINE_RANDOM_N=150 ENGINE_RANDOM_SEED=$seed python -m pytest tests/test_gpu_parity.py -q -k random_engine 2>&1 | tail -3 | sed "s/^/engine seed $seed: /"
  LANE_RANDOM_N=120 LANE_RANDOM_SEED=$seed LANE_ROWS_RANDOM_N=120 LANE_ROWS_RANDOM_SEED=$seed python -m pytest tests/test_gpu_parity.py -q -k random_shapes 2>&1 | tail -3 | sed "s/^/lane seed $seed: /"
done
} | tee gpurun_out/r3_random_sweeps.txt
