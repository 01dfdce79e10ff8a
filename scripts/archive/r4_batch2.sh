#!/bin/bash
# round 4, second GPU pass: the whole GPU suite on the new sources; lane-kernel variants A/B; WRITE_SIZE with and without the
# amplitude-gradient atomics; deterministic mode on the default scaler; the simulated multi-GPU shards
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b2; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --no-header 2>&1 | tail -25 ) > $O/pytest.log 2>&1
cat $O/pytest.log
SKIP_TESTS=1 bash scripts/r4_lane_ab.sh r4b r4b_acc1 r4b_pairs 2>&1 | tee $O/lane_ab.log
# WRITE_SIZE: the shipped kernel against the diagnostic build without the dz_f atomics (wrong gradients)
for v in libcareless_hip exp_r4c_nodzf; do
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
# deterministic mode on the default scaler
for det in 0 1; do
  for WL in mono_10M_cli_default_20x10_S1 mono_10M_studentt_posenc_20x10_S8 mono_10M_studentt_posenc_5x64_S8; do
    CARELESS_HIP_DETERMINISTIC=$det timeout 600 python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline > $O/det_$det.json 2> $O/det_$det.err || tail -3 $O/det_$det.err
    python3 -c "
import json,sys
d=json.loads(open('$O/det_$det.json').read().strip().splitlines()[-1]); r=d['roofline']
print('DET=$det %-40s ms/step %.4f kernel ms %.4f frac %.4f  %s' % ('$WL', d['ms_per_step'], r['kernel_ms'], r['frac'], r['kernel'].split(' (cl_')[0]))"
  done
done 2>&1 | tee $O/det.log
bash scripts/r4_sim_world.sh 2>&1 | tee $O/sim_world.log
