#!/bin/bash
# round 3, second GPU pass: the whole GPU suite on the extended kernels, the Student-T division A/B on the lane kernel, and the
# two-rank rehearsals (gloo, both ranks on this one GPU) of the bench's shared host data + extra-configuration protocol
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r3b_pytest.txt
{
for lib in "" careless_amd/lib/exp_slowdiv.so; do
  CARELESS_HIP_LIB=$lib SHAPES="10:5:1,10:5:8,10:21:8" python scripts/lane_shapes.py 2>&1 | grep -v Warning | sed "s|\$| lib=$lib|"
done
} | tee gpurun_out/r3b_fastdiv.txt
python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --nobs 2000000 --extra dw_50M_normal_5x64_S1 --extra-nobs 4000000 --no-cpu-baseline > gpurun_out/r3b_rehearsal_gloo2.json 2> gpurun_out/r3b_rehearsal_gloo2.err
echo "rehearsal rc=$?"; tail -c 1500 gpurun_out/r3b_rehearsal_gloo2.json; tail -5 gpurun_out/r3b_rehearsal_gloo2.err
GPUS="1" STEPS=10 bash scripts/scale_curve.sh gpurun_out/r3b_scale
