#!/bin/bash
# copy what scripts/r3_profiles.sh left under gpurun_out/r3p into profiles/ (tracked) and rebuild profiles/traffic.json
rm -f profiles/r3_kernel_stats_*.csv profiles/r3_bench_*.json profiles/r3_pmc_*.txt
for f in gpurun_out/r3p/kernel_stats_*.csv; do cp $f profiles/r3_$(basename $f); done
for f in gpurun_out/r3p/bench_*.json; do tail -1 $f > profiles/r3_$(basename $f); done
cp gpurun_out/r3p/summary.txt profiles/r3_profiles_summary.txt; cp gpurun_out/r3p/sources.txt profiles/r3_sources.txt
args=""
for f in gpurun_out/r3p/pmc_*.txt; do wl=$(basename $f .txt); wl=${wl#pmc_}; cp $f profiles/r3_pmc_$wl.txt; args="$args $wl=profiles/r3_pmc_$wl.txt"; done
python scripts/traffic_json.py $args | grep -E "hbm_bytes|sources"
echo "library sources now: $(python -c 'from careless_amd.build import source_hash; print(source_hash())')"
