cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/smallcli; mkdir -p $out
for n in 1000000 250000; do
rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_$n -o t -- python3 bench.py --workload mono_10M_cli_default_20x10_S1 --nobs $n --steps 50 --warmup 5 --no-cpu-baseline > $out/b_$n.json 2> $out/b_$n.err
f=$(find $out/p_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] || { echo "no kernel_stats.csv (the profiled command failed)"; continue 2>/dev/null || exit 1; }
echo "== nobs=$n: $(python3 -c "import json;d=json.loads(open('$out/b_$n.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],4),'ms/step, kernel',round(d['roofline']['kernel_ms'],4))")"
cut -d, -f1,2,4 $f | sed 's/"void at::native::[a-z_]*<[0-9, ]*at::native::\([A-Za-z]*\)[^"]*"/"\1"/' | cut -c1-90 | sed -n 2,10p
python3 bench.py --workload mono_10M_cli_default_20x10_S1 --nobs $n --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('outside profiler', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
python3 scripts/host_overhead.py 2>/dev/null | tail -2
