# usage: bash scripts/run_wl.sh WORKLOAD variant...   (bench one workload per experimental library, 2 rounds)
wl=$1; shift
for r in 1 2; do for v in "$@"; do
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$v.so python bench.py --workload $wl --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('exp_$v', d['config']['workload'], '%.3f ms'%d['ms_per_step'], '%.4g refl/s'%d['value'], 'loss', d['config']['final_loss'])"
done; done
