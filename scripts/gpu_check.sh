#!/bin/bash
# quick GPU loop: parity tests, both bench workloads (no CPU baseline), phase stamps
python -m pytest tests -m gpu -q -x --no-header 2>&1 | tail -2
for w in mono_1M_normal_5x64_S1 mono_10M_studentt_posenc_5x64_S8; do
python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '%.3e refl/s'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'])"
done
if [ "$1" == "stamps" ]; then python scripts/stamps.py 2>&1 | tail -17; fi
