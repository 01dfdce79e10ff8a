#!/bin/bash
# gpurun helper (round 4): rank 0's shard of the multi-GPU configurations BASELINE.json defines, run on ONE device with the all-reduce
# call in place (bench.py --sim-world W --force-dist): per-rank step time, fused-kernel time, and what is left around it.
#   cfg4 @ 4 ranks (Laue, by harmonic groups), cfg5 @ 8 ranks (double-Wilson, row split; one- and two-piece message),
#   the headline and the CLI-default workloads @ 8
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_sim; mkdir -p $O
run() {   # tag, workload, world, extra env
  env $4 timeout 900 python bench.py --workload $2 --sim-world $3 --force-dist --steps 30 --warmup 5 --no-cpu-baseline > $O/$1.json 2> $O/$1.err || tail -5 $O/$1.err
  python - "$1" "$O/$1.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-34s ms/step %.4f  kernel ms %.4f  around the kernel %.4f  frac %.4f  %s  %s" % (sys.argv[1], d["ms_per_step"], r["kernel_ms"], d["ms_per_step"] - r["kernel_ms"], r["frac"], d["config"]["parallelism"], r["kernel"].split(" (")[0]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2; do
run laue_5M_w1 laue_5M_normal_5x64_S1 1 X=0
run laue_5M_w4 laue_5M_normal_5x64_S1 4 X=0
run laue_5M_w4_two_piece laue_5M_normal_5x64_S1 4 CARELESS_HIP_SPLIT_MESSAGE=1
run dw_50M_w8 dw_50M_normal_5x64_S1 8 X=0
run dw_50M_w8_two_piece dw_50M_normal_5x64_S1 8 CARELESS_HIP_SPLIT_MESSAGE=1
run headline_w8 mono_10M_studentt_posenc_5x64_S8 8 X=0
run headline_w8_rows mono_10M_studentt_posenc_5x64_S8 8 CARELESS_HIP_OWNER_SHARD=0
run headline_w8_rows_two_piece mono_10M_studentt_posenc_5x64_S8 8 "CARELESS_HIP_OWNER_SHARD=0 CARELESS_HIP_SPLIT_MESSAGE=1"
run cli_default_w8 mono_10M_cli_default_20x10_S1 8 X=0
done
