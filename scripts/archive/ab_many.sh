# several experimental libraries on the two 64-wide bench workloads, one device: bash scripts/ab_many.sh variant...
for r in 1 2; do for v in "$@"; do for w in mono_1M_normal_5x64_S1 mono_10M_studentt_posenc_5x64_S8; do
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$v.so python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('exp_$v', d['config']['workload'][:8], '%.3f ms'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'])"
done; done; done
