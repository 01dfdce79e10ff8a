# fixed cost of one fused-kernel launch: T(1 tile per workgroup) and T(2 tiles per workgroup) -> fixed = 2 T1 - T2
# usage: bash scripts/fixed_cost.sh [exp variant ...]
for v in "${@:-default}"; do
if [ "$v" != default ]; then export CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$v.so; fi
for n in 32768 65536; do
python bench.py --nobs $n --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v nobs', d['config']['n_obs'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'step_ms %.4f'%d['ms_per_step'])"
done; done
