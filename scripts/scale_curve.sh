#!/bin/bash
# The north-star curve: reflections/s per ELBO step of the headline workload (10 M observations) on 1, 2, 4 and 8 GPUs of ONE node,
# strong scaling, back to back.  Every point is a fresh `python bench.py --gpus N` (the launcher starts N fresh rank processes and
# never touches the GPU itself: nothing is exec'ed after GPU initialisation); with 4 / 8 ranks the line also carries the Laue /
# double-Wilson configuration BASELINE.json quotes there (`extra_configs`) and the launcher exits non-zero if that did not run.
# Then, at every N >= 2 that ran: the same headline step with the reflection-owner split (CARELESS_HIP_OWNER_SHARD=1; the row split is
# the default until this has run once on a node, engine.py) and, at the largest N, the row split's message in two pieces
# (CARELESS_HIP_SPLIT_MESSAGE=1) -- so that ONE run on a node answers the three open questions of DESIGN 5.2.
#   bash scripts/scale_curve.sh [out_dir]        GPUS="1 2 4 8" STEPS=20 WARMUP=3
out=${1:-gpurun_out/scale}; mkdir -p "$out"
ngpu=$(python -c 'import torch; print(torch.cuda.device_count())')
rc_all=0
for n in ${GPUS:-1 2 4 8}; do
  if [ "$n" -gt "$ngpu" ]; then echo "SCALE n_gpus=$n skipped: $ngpu GPU(s) visible" | tee -a "$out/summary.txt"; continue; fi
  python bench.py --gpus "$n" --steps "${STEPS:-20}" --warmup "${WARMUP:-3}" $([ "$n" -gt 1 ] && echo --no-cpu-baseline) > "$out/bench_n$n.json" 2> "$out/bench_n$n.err"
  rc=$?; [ $rc -ne 0 ] && rc_all=$rc
  python - "$out/bench_n$n.json" "$n" "$rc" <<'PY' | tee -a "$out/summary.txt"
import json, sys
path, n, rc = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    ex = "".join("  [%s: %s]" % (k, ("%.3e refl/s %.3f ms" % (v["value"], v["ms_per_step"])) if "value" in v else str(v)) for k, v in d.get("extra_configs", {}).items())
    print("SCALE n_gpus=%s rc=%s value=%.4e refl/s ms_per_step=%.3f kernel_ms=%.3f frac=%.3f rss_gib=%s%s" % (
        n, rc, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("host_peak_rss_gib_per_rank"), ex))
except Exception as e:
    print("SCALE n_gpus=%s rc=%s FAILED: %r" % (n, rc, e))
PY
  last=$n
done
summ() {
python - "$1" "$2" <<'PY' | tee -a "$out/summary.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("SCALE_AB %s value=%.4e refl/s ms_per_step=%.3f kernel_ms=%.3f parallelism=%s" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"].get("parallelism")))
except Exception as e:
    print("SCALE_AB %s FAILED: %r" % (sys.argv[2], e))
PY
}
for n in ${GPUS:-1 2 4 8}; do
  if [ "$n" -lt 2 ] || [ "$n" -gt "$ngpu" ]; then continue; fi
  CARELESS_HIP_OWNER_SHARD=1 python bench.py --gpus "$n" --steps "${STEPS:-20}" --warmup "${WARMUP:-3}" --no-cpu-baseline --extra none > "$out/bench_owner_n$n.json" 2> "$out/bench_owner_n$n.err"
  summ "$out/bench_owner_n$n.json" "owner_split n_gpus=$n"
done
if [ -n "$last" ] && [ "$last" -ge 2 ]; then
  CARELESS_HIP_SPLIT_MESSAGE=1 python bench.py --gpus "$last" --steps "${STEPS:-20}" --warmup "${WARMUP:-3}" --no-cpu-baseline --extra none > "$out/bench_two_piece_n$last.json" 2> "$out/bench_two_piece_n$last.err"
  summ "$out/bench_two_piece_n$last.json" "row_split_two_piece_message n_gpus=$last"
fi
exit $rc_all
