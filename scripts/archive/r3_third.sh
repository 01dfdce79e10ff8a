#!/bin/bash
# round 3, third GPU pass: chunked launches, deterministic mode, the exact gradient gate -- then the whole GPU suite
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "several_launches or deterministic or full_size_properties" 2>&1 | tail -15 | tee gpurun_out/r3c_new.txt
python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee gpurun_out/r3c_pytest.txt
