#!/bin/bash
# round 4, fourth GPU pass: suite; per-image-layer flush as stores (A/B against the old unit); deterministic mode with slot stores;
# kernel trace of the width-128 workload
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b4; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -25 ) > $O/pytest.log 2>&1
cat $O/pytest.log
line() {
python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-52s ms/step %.4f kernel ms %.4f frac %.4f step frac %.4f  %s" % (sys.argv[1], d["ms_per_step"], r["kernel_ms"], r["frac"], r["frac_on_step_time"], r["kernel"].split(" (cl_")[0]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2; do
for v in exp_r4e_imgl_old libcareless_hip; do
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/$v.so timeout 600 python bench.py --workload mono_10M_studentt_posenc_4x64_img1_S8 --steps 20 --warmup 3 --no-cpu-baseline > $O/imgl_$v.json 2> $O/imgl_$v.err || tail -3 $O/imgl_$v.err
  line "$v image layers" $O/imgl_$v.json
done
done 2>&1 | tee $O/imgl_ab.log
for det in 0 1; do
  for WL in mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_20x10_S8 mono_10M_studentt_posenc_5x64_S8; do
    CARELESS_HIP_DETERMINISTIC=$det timeout 600 python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline > $O/det_$det.json 2> $O/det_$det.err || tail -3 $O/det_$det.err
    line "DET=$det $WL" $O/det_$det.json
  done
done 2>&1 | tee $O/det.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_wide -o t -- python3 bench.py --workload mono_2M_studentt_3x128_S4 --steps 10 --warmup 3 --no-cpu-baseline > $O/wide_bench.json 2> $O/wide_bench.err
f=$(find $O/prof_wide -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/wide_kernel_stats.csv && head -25 $O/wide_kernel_stats.csv | cut -c1-160
rm -rf $O/prof_wide
