#!/bin/bash
# Round 6: per-image layers at other depths than the default on the lane kernel (the per-depth units' NI instances) -- first run on the hardware:
# the new parity / routing / repeatability cases, then `--mlp-layers 10 --image-layers 2` at 10 M observations with the instances on and off.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
out=gpurun_out/r6; mkdir -p $out
{
python3 -c "from careless_amd.build import source_hash; print('sources', source_hash())"
timeout 1500 python3 -m pytest tests/test_routing.py tests/test_gpu_parity.py tests/test_lane_repeat.py -m gpu -q -x -k "image_layers or imgl" 2>&1 | tail -8
for on in 1 0; do
  for wl in mono_10M_10x10_img2_S1; do
    echo "# $wl CARELESS_HIP_LANE_DEPTHS=$on"
    CARELESS_HIP_LANE_DEPTHS=$on python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%.4g refl/s  %.3f ms/step  kernel %.3f ms  frac %.3f  %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('kernel')))"
  done
done
for on in 1 0; do
  echo "# envelope, two per-image layers, CARELESS_HIP_LANE_DEPTHS=$on"
  CARELESS_HIP_LANE_DEPTHS=$on IMGL=2 LS=5,10,16 WS=7,10 DS=5,21 SS=1,8 python3 scripts/envelope.py
done
} 2>&1 | tee $out/imgl_depth.txt
