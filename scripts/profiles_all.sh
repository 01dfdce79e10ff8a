#!/bin/bash
# The round's evidence on the current sources, one gpurun: for every workload the bench line of `bench.py --workload W` and a kernel-trace
# summary of the same command run again under rocprofv3 (--kernel-trace --stats, program directly after --); then the PMC passes (separate --pmc runs,
# never combined with trace domains) of EVERY workload -- per step: summed over the dominant kernels' launches of a step (one launch for the
# fused kernels, the GEMM launches of the layer-by-layer path) -- and profiles/traffic.json from them (with the source hash).
#   WLS="..." PMC_WLS="..." bash scripts/profiles_all.sh
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
out=gpurun_out/rprof; mkdir -p $out
python3 -c "from careless_amd.build import source_hash; print(source_hash())" > $out/sources.txt
ALL="mono_10M_10x10_S1 mono_10M_10x10_img2_S1 mono_10M_20x10_img3_S1 mono_10M_24x10_S1 mono_10M_studentt_posenc_5x64_S8 mono_1M_normal_5x64_S1 laue_5M_normal_5x64_S1 mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_20x10_S8 mono_10M_studentt_posenc4_20x10_S8 mono_10M_studentt_posenc_4x64_img1_S8 mono_2M_studentt_3x128_S4 dw_50M_normal_5x64_S1 mono_10M_20x10_img2_S1 laue_5M_normal_20x10_S1 laue_5M_normal_20x10_img2_S1 dw_10M_normal_20x10_S1 mono_10M_studentt_posenc_20x10_img2_S8"
WLS=${WLS:-$ALL}
for wl in $WLS; do
  # the bench line from a run of its own (round 6: a line taken under the tracer is profiler-perturbed), then the same command under the tracer
  python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_$wl.json 2> $out/bench_$wl.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$wl -o t -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $out/traced_$wl.json 2> $out/traced_$wl.err
  f=$(find $out/prof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] || { echo "PROF $wl: no kernel_stats.csv (the profiled command failed)"; tail -3 $out/bench_$wl.err; continue; }
  cp $f $out/kernel_stats_$wl.csv
  rm -rf $out/prof_$wl
  python3 - $out/bench_$wl.json $out/kernel_stats_$wl.csv <<'PY'
import sys, json, csv
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    rows = list(csv.DictReader(open(sys.argv[2])))
    top = rows[0]
    print("PROF %-40s %.4g refl/s %.3f ms/step | live kernel %.3f ms frac %.3f | rocprof top: %s calls %s avg %.3f ms (%s%%) | build %s" % (
        d["config"]["workload"], d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], top["Name"][:60], top["Calls"], float(top["AverageNs"]) / 1e6, top["Percentage"], d.get("build")))
except Exception as e:
    print("PROF", sys.argv[1], "FAILED", e)
PY
done | tee $out/summary.txt
PMC_WLS=${PMC_WLS:-$ALL}
for wl in $PMC_WLS; do
  bash scripts/pmc_passes_step.sh $wl > $out/pmc_$wl.txt 2>&1
  grep -E "^[A-D] " $out/pmc_$wl.txt | head -40
done
rm -rf gpurun_out/pmc?_* gpurun_out/pmc?.log
