#!/bin/bash
# A one-off host stall of 35 - 55 ms inside the timed region of some short runs (rank 0's shard of a simulated 8-rank job, row split,
# exactly 40 steps: 0.35 -> 1.26 ms per step) was a generation-2 pass of Python's garbage collector.  bench.py now collects before the
# timed region and keeps the collector off inside it; BENCH_KEEP_GC=1 leaves it on (the collection in front alone moves the pass away)
cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; }
W=mono_10M_cli_default_20x10_S1
for st in 40 40 20 80; do
BENCH_KEEP_GC=1 CARELESS_HIP_OWNER_SHARD=0 python3 bench.py --workload $W --steps $st --warmup 5 --no-cpu-baseline --sim-world 8 --force-dist 2>/dev/null | line "rows  steps=$st gc on :"
CARELESS_HIP_OWNER_SHARD=0 python3 bench.py --workload $W --steps $st --warmup 5 --no-cpu-baseline --sim-world 8 --force-dist 2>/dev/null | line "rows  steps=$st gc off:"
done
