#!/bin/bash
# round-2 evidence for the lane-per-observation kernel: whole -m gpu suite, the CLI-default bench line with and without the kernel
# (CARELESS_HIP_LANE=0 = elbo_narrow.hip), kernel-trace stats and the PMC passes of the CLI-default workload
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2l; mkdir -p $O
python -m pytest tests -m gpu -q -x --no-header > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
wl=mono_10M_cli_default_20x10_S1
for v in 1 0; do CARELESS_HIP_LANE=$v python bench.py --workload $wl --no-cpu-baseline > $O/bench_${wl}_lane$v.json 2> $O/bench_lane$v.err; tail -c 700 $O/bench_${wl}_lane$v.json | head -c 400; echo; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_${wl}_under_rocprof.json 2> $O/bench_rocprof.err
f=$(find $O/stats_$wl -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$wl.csv; cut -d, -f1-4 $f | cut -c1-110 | head -4; rm -rf $O/stats_$wl
bash scripts/pmc_passes.sh $wl > $O/pmc_$wl.txt 2>&1; tail -22 $O/pmc_$wl.txt
