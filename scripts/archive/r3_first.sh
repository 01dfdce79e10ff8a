#!/bin/bash
# round 3, first GPU pass: parity of the extended lane kernel, then step times of the shapes it now takes (A/B against the
# kernels those shapes ran on before)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "lane or narrow or cli_default" 2>&1 | tail -15 > gpurun_out/r3a_pytest.txt
cat gpurun_out/r3a_pytest.txt
{
python scripts/lane_shapes.py
CARELESS_HIP_LANE=0 SHAPES="10:5:1,10:5:8,10:21:1,10:21:8,10:31:1,10:5:12" python scripts/lane_shapes.py
CARELESS_HIP_NARROW_W4=1 SHAPES="13:5:1,15:15:1,15:15:8" python scripts/lane_shapes.py
SHAPES="15:15:8" python scripts/lane_shapes.py
} 2>&1 | grep -v Warning | tee gpurun_out/r3a_shapes.txt
bash scripts/bench_configs.sh 2>&1 | tee gpurun_out/r3a_bench_configs.txt
python bench.py --workload mono_10M_studentt_posenc_20x10_S8 --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r3a_bench_posenc_20x10.json
