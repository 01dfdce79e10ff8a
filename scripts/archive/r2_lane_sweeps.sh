#!/bin/bash
# evidence for DESIGN 4.11: the lane-per-observation kernel against elbo_narrow.hip (CARELESS_HIP_LANE=0) over MC samples and scaler shapes,
# 4 M observations, one device
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in 1 0; do
  CARELESS_HIP_LANE=$v SAMPLES=1,2,3,4,6,8 timeout 600 python scripts/narrow_samples.py 2>&1 | grep mono
  CARELESS_HIP_LANE=$v timeout 900 python scripts/narrow_shapes.py "20,10,5 20,10,12 20,8,8 20,8,15 20,6,6 20,5,5 20,4,4 12,10,5 20,12,5 20,13,13" 2>&1 | grep mono
  CARELESS_HIP_LANE=$v timeout 300 python scripts/laue_default_scaler.py 2>&1 | grep laue
done
} | tee gpurun_out/lane_sweeps.txt
