#!/bin/bash
# Round 6, second call of the closing evidence (after scripts/store_profiles.sh r6 has written profiles/traffic.json for the final sources):
# the default `python bench.py` run -- its roofline.traffic is then measured on the same sources (`traffic_from.stale` false) -- and a soak of
# the lane kernel's instances (CLI default, dZ0-storing, per-image layers, another depth, a chain of lane blocks): 60 launches at 10 M observations
# on one engine, every output against the first; then a longer randomized parity sweep over round 6's routes on other seeds.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
mkdir -p gpurun_out/r6
timeout 1500 python3 bench.py > gpurun_out/r6/bench_default_run.json 2> gpurun_out/r6/bench_default_run.err; tail -c 1200 gpurun_out/r6/bench_default_run.json
for cfg in cli_default peeled_dZ0_out image_layers2_peeled_d21 depth10_10x10 chain_24x10 image_layers2_depth10 headline_5x64 narrow_12x12; do
  timeout 900 python3 scripts/probe/lane_defect_probe.py --config $cfg --runs 60 --N 10000000 --images 9973 --same-engine --tag soak_$cfg 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
print(r['tag'], r['kernel'], 'launches', r['runs'], 'that differ from the first:', r['n_bad_runs'], 'distinct NLL', r['distinct_nll'], 'max gradient difference / max-norm', max([p['grad_maxdiff_rel'] for p in r['per_run']] or [0]))"
done | tee gpurun_out/r6/soak.txt
{
echo "Randomized sweep over round 6's routes on the final sources $(python3 -c 'from careless_amd.build import source_hash; print(source_hash())') (tests/test_gpu_parity.py: _random_engine_cases,"
echo "_random_lane_depth_cases: widths 5 .. 12, depths 2 .. 40, 1 .. 40 columns; _random_lane_image_layer_cases at the default and at other depths), seeds other than the suite's:"
for seed in 4 29; do
  ENGINE_RANDOM_SEED=$seed ENGINE_RANDOM_N=60 LANE_DEPTH_RANDOM_N=120 LANE_IMGL_RANDOM_N=40 LANE_IMGL_DEPTH_RANDOM_N=80 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -k random_engine 2>&1 | tail -2 | tr '\n' ' '
  echo " (seed $seed)"
done
} | tee gpurun_out/r6/random_sweep.txt
