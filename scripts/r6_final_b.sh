#!/bin/bash
# Round 6, second call of the closing evidence (after scripts/store_profiles.sh r6 has written profiles/traffic.json for the final sources):
# the default `python bench.py` run -- its roofline.traffic is then measured on the same sources (`traffic_from.stale` false) -- and a soak of
# the CLI-default kernel: 60 launches at 10 M observations on one engine, every output against the first.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
mkdir -p gpurun_out/r6
timeout 1500 python3 bench.py > gpurun_out/r6/bench_default_run.json 2> gpurun_out/r6/bench_default_run.err; tail -c 1200 gpurun_out/r6/bench_default_run.json
for cfg in cli_default peeled_dZ0_out image_layers2_peeled_d21; do
  timeout 900 python3 scripts/probe/lane_defect_probe.py --config $cfg --runs 60 --N 10000000 --images 9973 --same-engine --tag soak_$cfg 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
print(r['tag'], r['kernel'], 'launches', r['runs'], 'that differ from the first:', r['n_bad_runs'], 'distinct NLL', r['distinct_nll'], 'max gradient difference / max-norm', max([p['grad_maxdiff_rel'] for p in r['per_run']] or [0]))"
done | tee gpurun_out/r6/soak.txt
