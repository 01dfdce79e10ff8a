#!/bin/bash
# round-2 evidence: kernel-trace stats + PMC passes of the bench line and of the CLI-default workload, simulated 8-rank step,
# CPU baseline at the workload's full size
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2p; mkdir -p $O
for wl in mono_10M_studentt_posenc_5x64_S8 mono_10M_cli_default_20x10_S1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
  f=$(find $O/stats_$wl -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$wl.csv; cut -d, -f1-4 $f | cut -c1-110 | head -5; rm -rf $O/stats_$wl
  bash scripts/pmc_passes.sh $wl > $O/pmc_$wl.txt 2>&1; grep "^A\|^B\|^C\|^D" $O/pmc_$wl.txt | head -30
done
for W in 1 2 4 8; do python bench.py --sim-world $W --force-dist --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('SIM world $W: %.3f ms/step, fused kernel %.3f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done
python bench.py --steps 5 --warmup 2 --cpu-full > $O/bench_cpu_full.json 2> $O/bench_cpu_full.err; tail -c 900 $O/bench_cpu_full.json
