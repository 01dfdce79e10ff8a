mkdir -p gpurun_out/r5b
for w in mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_5x64_S8; do
for n in 20000 100000 400000 1600000; do
python bench.py --workload $w --nobs $n --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{}); print('$w', $n, 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', r.get('kernel_ms'), 'keys', [k for k in d if 'ms' in k])
"
done; done | tee gpurun_out/r5b/small_steps.txt
