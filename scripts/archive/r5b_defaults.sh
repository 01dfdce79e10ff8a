mkdir -p gpurun_out/r5b
for w in laue_5M_normal_20x10_S1 dw_10M_normal_20x10_S1 mono_10M_20x10_img2_S1 mono_10M_cli_default_20x10_S1; do
python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d.get('roofline',{}); print('$w', 'ms/step', round(d['ms_per_step'],4), 'kernel', r.get('kernel'), 'kernel_ms', r.get('kernel_ms'), 'frac', r.get('frac'), 'on_step', r.get('achieved_on_step_time'))
"
done | tee gpurun_out/r5b/defaults.txt
