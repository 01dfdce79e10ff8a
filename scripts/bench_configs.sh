#!/bin/bash
# every BASELINE.json configuration (and two extra scaler geometries) on ONE GPU; one summary line each
for wl in mono_1M_normal_5x64_S1 mono_10M_studentt_posenc_5x64_S8 laue_5M_normal_5x64_S1 dw_50M_normal_5x64_S1 mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_4x64_img1_S8; do
  python bench.py --workload $wl --steps ${STEPS:-15} --warmup 3 --no-cpu-baseline 2>/tmp/bench_err.log | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    r = d['roofline']
    print('CFG %-40s %.4g refl/s  %.3f ms/step  kernel %.3f ms  mfma_frac %.3f  loss_finite %s' % (d['config']['workload'], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], d['config']['loss_finite']))
except Exception as e:
    print('CFG $wl FAILED', e); print(open('/tmp/bench_err.log').read()[-1500:])
"
done
