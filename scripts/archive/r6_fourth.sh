#!/bin/bash
# Round 6, fourth GPU call: the lane kernel compiled per depth (--mlp-layers 2 .. 19 at widths 7 .. 10), the frozen step again, the envelope
# table with the depth instances on and off, the eight-rank gloo rehearsal.
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_frozen_scaler.py tests/test_routing.py tests/test_lane_repeat.py tests/test_gpu_parity.py -m gpu -q -k "frozen or sorted or routing or depth or narrow or mlp9x7 or design" ) > gpurun_out/r6_fourth_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/r6_fourth_tests.txt
tail -15 gpurun_out/r6_fourth_tests.txt
: > gpurun_out/r6_frozen_step.jsonl
for W in mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_5x64_S8 laue_5M_normal_5x64_S1 mono_10M_20x10_img2_S1 dw_10M_normal_20x10_S1; do
  timeout 600 python3 scripts/frozen_bench.py $W 2>/dev/null | tail -1 >> gpurun_out/r6_frozen_step.jsonl
done
cut -c1-260 gpurun_out/r6_frozen_step.jsonl
N=4000000 LS=2,5,8,10,12,16,19,20 WS=7,8,10 DS=5,12 SS=1,8 timeout 1500 python3 scripts/envelope.py > gpurun_out/r6_envelope_depths.txt 2>&1
CARELESS_HIP_LANE_DEPTHS=0 N=4000000 LS=2,5,8,10,12,16,19,20 WS=7,8,10 DS=5,12 SS=1,8 timeout 1500 python3 scripts/envelope.py > gpurun_out/r6_envelope_depths_off.txt 2>&1
cat gpurun_out/r6_envelope_depths.txt
bash scripts/r6_rehearsal.sh
