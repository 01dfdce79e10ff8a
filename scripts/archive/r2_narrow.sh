#!/bin/bash
# narrow kernel: parity cases first, then A/B against the eight-wave instance on the CLI-default workload
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --no-header -k "cli_default or mlp9x7 or mlp7x12 or mlp5x13 or trajectory or three_obs or rank_shards or full_size" > gpurun_out/r2n/pytest.log 2>&1; tail -15 gpurun_out/r2n/pytest.log
for v in 1 0; do
CARELESS_HIP_NARROW=$v timeout 600 python bench.py --workload mono_10M_cli_default_20x10_S1 --steps 15 --warmup 3 --no-cpu-baseline 2>gpurun_out/r2n/err_$v.log | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('NARROW=$v', '%.3e refl/s'%d['value'], '%.3f ms'%d['ms_per_step'], 'kern %.3f'%r['kernel_ms'], 'frac %.3f'%r['frac'], d['config']['final_loss'])"
done
