#!/bin/bash
# experiment: the packed compilation units (row_map + per-row noise key) on a plain mono problem -- what an owner-computes shard would cost
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/exp_packed; mkdir -p $out
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['kernel'])"; }
for wl in mono_10M_studentt_posenc_5x64_S8 mono_10M_cli_default_20x10_S1 mono_1M_normal_5x64_S1; do
  for pk in 0 1; do
    CARELESS_HIP_EXPERIMENT_PACKED=$pk python3 bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline 2>$out/err_$wl_$pk.txt | line "FULL $wl packed=$pk"
    CARELESS_HIP_EXPERIMENT_PACKED=$pk python3 bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --force-dist --sim-world 8 2>>$out/err_$wl_$pk.txt | line "SIM8 $wl packed=$pk"
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_sim8 -o t -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --force-dist --sim-world 8 > $out/sim8.json 2> $out/sim8.err
f=$(find $out/prof_sim8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] || { echo "no kernel_stats.csv (the profiled command failed)"; continue 2>/dev/null || exit 1; }; cut -d, -f1-4 $f | cut -c1-150 | head -14
