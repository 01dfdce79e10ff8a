set -x
for W in 2 4 8; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --force-dist --sim-world $W 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SIM', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FULL', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
