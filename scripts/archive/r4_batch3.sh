#!/bin/bash
# round 4, third GPU pass: parity of the wide path with the recomputed first layer + fused head; its A/B; lane-kernel variants; WRITE_SIZE
# with and without the amplitude-gradient atomics; deterministic mode with the 16-lane reduce
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b3; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -25 ) > $O/pytest.log 2>&1
cat $O/pytest.log
line() {
python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-44s ms/step %.4f kernel ms %.4f frac %.4f step frac %.4f  %s" % (sys.argv[1], d["ms_per_step"], r["kernel_ms"], r["frac"], r["frac_on_step_time"], r["kernel"].split(" (cl_")[0]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2; do
for pre in 0 1; do
  CARELESS_HIP_WIDE_PRE=$pre timeout 600 python bench.py --workload mono_2M_studentt_3x128_S4 --steps 20 --warmup 3 --no-cpu-baseline > $O/wide_$pre.json 2> $O/wide_$pre.err || tail -3 $O/wide_$pre.err
  line "WIDE_PRE=$pre mono_2M_studentt_3x128_S4" $O/wide_$pre.json
done
done 2>&1 | tee $O/wide_ab.log
SKIP_TESTS=1 bash scripts/r4_lane_ab.sh r4d r4d_acc1 r4d_pairs 2>&1 | tee $O/lane_ab.log
for v in libcareless_hip exp_r4d_nodzf; do
  rm -rf gpurun_out/pmcW_$v
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/$v.so rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/pmcW_$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload mono_10M_cli_default_20x10_S1 > $O/pmcW_$v.log 2>&1
  python3 - $v <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/pmcW_{sys.argv[1]}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "elbo_lane" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("WRITE", sys.argv[1], {k: (len(v), sum(v) / len(v)) for k, v in acc.items()})
PY
  rm -rf gpurun_out/pmcW_$v
done 2>&1 | tee $O/write_size.log
for det in 0 1; do
  for WL in mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_20x10_S8 laue_5M_normal_5x64_S1; do
    [ "$WL" = laue_5M_normal_5x64_S1 ] && [ $det = 1 ] && continue
    CARELESS_HIP_DETERMINISTIC=$det timeout 600 python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline > $O/det_$det.json 2> $O/det_$det.err || tail -3 $O/det_$det.err
    line "DET=$det $WL" $O/det_$det.json
  done
done 2>&1 | tee $O/det.log
