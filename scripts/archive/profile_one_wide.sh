cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r3p; mkdir -p $out; wl=mono_2M_studentt_3x128_S4
rm -rf $out/prof_$wl
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$wl -o t -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_$wl.json 2> $out/bench_$wl.err
f=$(find $out/prof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] || { echo failed; exit 1; }
cp $f $out/kernel_stats_$wl.csv; rm -rf $out/prof_$wl
python3 -c "
import json,csv; d=json.loads(open('$out/bench_$wl.json').read().strip().splitlines()[-1]); r=d['roofline']; rows=list(csv.DictReader(open('$out/kernel_stats_$wl.csv'))); top=rows[0]
print('PROF %-40s %.4g refl/s %.3f ms/step | live kernel %.3f ms frac %.3f | rocprof top: %s calls %s avg %.3f ms (%s%%) | build %s' % (d['config']['workload'], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], top['Name'][:60], top['Calls'], float(top['AverageNs'])/1e6, top['Percentage'], d.get('build')))"
