#!/bin/bash
# owner-mode vs row-split: simulated rank-0 shard of an 8-rank job (one GPU), headline + CLI default
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/exp_owner; mkdir -p $out
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), d['config']['parallelism'], d['roofline']['kernel'])"; }
for wl in mono_10M_studentt_posenc_5x64_S8 mono_10M_cli_default_20x10_S1 mono_1M_normal_5x64_S1; do
  for W in 2 4 8; do
  for own in 0 1; do
    CARELESS_HIP_OWNER_SHARD=$own python3 bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --force-dist --sim-world $W 2>>$out/err.txt | line "SIM$W $wl owner=$own"
  done
  done
done
CARELESS_HIP_OWNER_SHARD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_sim8 -o t -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --force-dist --sim-world 8 > $out/sim8.json 2> $out/sim8.err
f=$(find $out/prof_sim8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] || { echo "no kernel_stats.csv (the profiled command failed)"; continue 2>/dev/null || exit 1; }; cut -d, -f1-4 $f | cut -c1-110 | head -14
