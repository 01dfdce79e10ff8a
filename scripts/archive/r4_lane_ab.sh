#!/bin/bash
# gpurun helper (round 4): parity of everything the lane kernel runs, then A/B of library variants (careless_amd/lib/exp_NAME.so)
# on the two default-scaler workloads and on rank 0's shard of a simulated 8-rank job, all on ONE device.
#   bash scripts/r4_lane_ab.sh NAME...          (env ROUNDS, default 2; SKIP_TESTS=1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lane
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x --no-header -k "cli_default or lane or narrow or trajectory or rank_shards or laue or ev11 or golden or owner" 2>&1 | tail -8
fi
line() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-28s" % sys.argv[1], "ms/step", round(d["ms_per_step"], 4), "kernel ms", round(d["roofline"].get("kernel_ms", 0), 4), "frac", round(d["roofline"]["frac"], 4), d["roofline"]["kernel"].split(" (")[0])
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for rep in $(seq 1 ${ROUNDS:-2}); do
for v in "$@"; do
  for WL in mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_20x10_S8; do
    CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$v.so timeout 600 python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/lane/m_$v.json 2> gpurun_out/lane/m_$v.err || tail -5 gpurun_out/lane/m_$v.err
    line "$v $WL" gpurun_out/lane/m_$v.json
  done
  CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$v.so timeout 600 python bench.py --workload mono_10M_cli_default_20x10_S1 --sim-world 8 --force-dist --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/lane/s_$v.json 2> gpurun_out/lane/s_$v.err || tail -5 gpurun_out/lane/s_$v.err
  line "$v SIM8 cli_default" gpurun_out/lane/s_$v.json
done
done
