#!/bin/bash
# gpurun helper (round 5): PMC passes over the split-bf16 probe's kernels (separate --pmc runs, nothing else traced)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
P="scripts/probe/split_bf16_probe 64 8"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/r5/pmcA -- $P > gpurun_out/r5/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d gpurun_out/r5/pmcB -- $P > gpurun_out/r5/pmcB.log 2>&1
python3 - <<'PY' | tee gpurun_out/r5/split_bf16_probe_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in "AB":
    for f in glob.glob(f"gpurun_out/r5/pmc{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        v = v[-3:] if len(v) >= 4 else v          # the timing launches (the first launch of a kernel is the small check problem)
        print("    %-28s launches %d  per launch %.4g" % (n, len(v), sum(v) / len(v)))
PY
