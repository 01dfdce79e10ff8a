#!/bin/bash
# Round 6: deterministic mode with per-image layers on the lane kernel's instances (one wave per image) -- first run on the hardware:
# parity + bit-for-bit repeatability cases, then what the mode costs on `--image-layers 2` at 10 M observations.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
out=gpurun_out/r6; mkdir -p $out
{
python3 -c "from careless_amd.build import source_hash; print('sources', source_hash())"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_lane_repeat.py -m gpu -q -x -k "deterministic or det_" 2>&1 | tail -8
for wl in mono_10M_20x10_img2_S1 mono_10M_10x10_img2_S1 laue_5M_normal_20x10_img2_S1; do
  for det in 0 1; do
    echo "# $wl CARELESS_HIP_DETERMINISTIC=$det"
    CARELESS_HIP_DETERMINISTIC=$det python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%.4g refl/s  %.3f ms/step  kernel %.3f ms  frac %.3f  %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('kernel')))"
  done
done
} 2>&1 | tee $out/imgl_det.txt
