#!/bin/bash
# Round 6: three per-image layers on the lane kernel (default depth; a unit compiled without -amdgpu-mfma-vgpr-form) -- first run on the
# hardware: parity / routing / repeatability / deterministic cases, then `--image-layers 3` at 10 M observations with the lane kernel on and off.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
out=gpurun_out/r6; mkdir -p $out
{
python3 -c "from careless_amd.build import source_hash; print('sources', source_hash())"
timeout 1500 python3 -m pytest tests/test_routing.py tests/test_gpu_parity.py tests/test_lane_repeat.py -m gpu -q -x -k "image_layers or imgl" 2>&1 | tail -8
for on in 1 0; do
  echo "# mono_10M_20x10_img3_S1 CARELESS_HIP_LANE=$on"
  CARELESS_HIP_LANE=$on python3 bench.py --workload mono_10M_20x10_img3_S1 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%.4g refl/s  %.3f ms/step  kernel %.3f ms  frac %.3f  %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('kernel')))"
done
} 2>&1 | tee $out/imgl3.txt
