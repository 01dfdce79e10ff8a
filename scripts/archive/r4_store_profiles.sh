#!/bin/bash
# copy what scripts/r4_profiles.sh left under gpurun_out/r4p into profiles/ (tracked) and rebuild profiles/traffic.json
for f in gpurun_out/r4p/kernel_stats_*.csv; do cp $f profiles/r4_$(basename $f); done
for f in gpurun_out/r4p/bench_*.json; do tail -1 $f > profiles/r4_$(basename $f); done
cp gpurun_out/r4p/summary.txt profiles/r4_profiles_summary.txt; cp gpurun_out/r4p/sources.txt profiles/r4_sources.txt
args=""
for f in gpurun_out/r4p/pmc_*.txt; do wl=$(basename $f .txt); wl=${wl#pmc_}; cp $f profiles/r4_pmc_$wl.txt; args="$args $wl=profiles/r4_pmc_$wl.txt"; done
python scripts/traffic_json.py $args | grep -E "hbm_bytes|sources"
echo "library sources now: $(python -c 'from careless_amd.build import source_hash; print(source_hash())')"
