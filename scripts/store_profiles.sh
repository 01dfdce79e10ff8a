#!/bin/bash
# copy what scripts/profiles_all.sh left under gpurun_out/rprof into profiles/ (tracked) under the round's tag and rebuild profiles/traffic.json
#   bash scripts/store_profiles.sh r5
tag=${1:?round tag, e.g. r5}
for f in gpurun_out/rprof/kernel_stats_*.csv; do cp $f profiles/${tag}_$(basename $f); done
for f in gpurun_out/rprof/bench_*.json; do tail -1 $f > profiles/${tag}_$(basename $f); done
cp gpurun_out/rprof/summary.txt profiles/${tag}_profiles_summary.txt; cp gpurun_out/rprof/sources.txt profiles/${tag}_sources.txt
args=""
for f in gpurun_out/rprof/pmc_*.txt; do wl=$(basename $f .txt); wl=${wl#pmc_}; cp $f profiles/${tag}_pmc_$wl.txt; args="$args $wl=profiles/${tag}_pmc_$wl.txt"; done
python scripts/traffic_json.py $args | grep -E "hbm_bytes|sources"
echo "library sources now: $(python -c 'from careless_amd.build import source_hash; print(source_hash())')"
