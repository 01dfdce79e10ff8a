#!/bin/bash
# Round 6, second GPU call: the repaired library -- (1) every lane instance's repeatability at 5 000 and 4 M rows, (2) the whole GPU suite,
# (3) round-5 library against the round-6 one on the workloads whose kernels changed their LeakyReLU (alternating, same box).
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
mkdir -p gpurun_out
( time timeout 2400 python3 -m pytest tests/test_lane_repeat.py -m gpu -q -x --durations=5 ) > gpurun_out/r6_lane_repeat.txt 2>&1
echo "lane_repeat rc=$?" >> gpurun_out/r6_lane_repeat.txt
tail -5 gpurun_out/r6_lane_repeat.txt
( time timeout 3000 python3 -m pytest tests -m gpu -q --deselect tests/test_lane_repeat.py ) > gpurun_out/r6_gpu_suite_a.txt 2>&1
echo "suite rc=$?" >> gpurun_out/r6_gpu_suite_a.txt
tail -5 gpurun_out/r6_gpu_suite_a.txt
: > gpurun_out/r6_ab_lrelu.jsonl
for rep in 1 2; do
for W in mono_10M_studentt_posenc_5x64_S8 mono_1M_normal_5x64_S1 laue_5M_normal_5x64_S1 mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_20x10_S8 laue_5M_normal_20x10_S1 mono_10M_20x10_img2_S1 mono_10M_studentt_posenc_20x10_img2_S8 mono_10M_studentt_posenc4_20x10_S8 mono_10M_10x10_S1 mono_10M_studentt_posenc_4x64_img1_S8; do
  for L in r5 r6; do
    if [ $L = r5 ]; then export CARELESS_HIP_LIB=$PWD/careless_amd/lib/variants/libcareless_hip_r5.so; else unset CARELESS_HIP_LIB; fi
    timeout 600 python3 bench.py --workload $W --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline()); print(json.dumps(dict(lib='$L', workload='$W', ms_per_step=r['ms_per_step'], kernel_ms=r.get('roofline', {}).get('kernel_ms'), frac=r['roofline']['frac'], kernel=r.get('config', {}).get('kernel'))))" >> gpurun_out/r6_ab_lrelu.jsonl
  done
done
done
unset CARELESS_HIP_LIB
cat gpurun_out/r6_ab_lrelu.jsonl | cut -c1-200
