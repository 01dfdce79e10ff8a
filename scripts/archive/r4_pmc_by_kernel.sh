#!/bin/bash
# usage: bash scripts/r4_pmc_by_kernel.sh WORKLOAD  -- two PMC passes, counters summed PER KERNEL NAME over 3 steps (where does each GEMM kernel's time go)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$1
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcKA_$T -- $B > gpurun_out/pmcKA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmcKB_$T -- $B > gpurun_out/pmcKB.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS --output-format csv -d gpurun_out/pmcKC_$T -- $B > gpurun_out/pmcKC.log 2>&1
python3 - $T <<'PY'
import csv, glob, collections, sys, re
T = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for d in ("KA", "KB", "KC"):
    for f in glob.glob(f"gpurun_out/pmc{d}_{T}/*/*counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            if not n.startswith(("wide_", "elbo_", "laue_")): continue
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
            if d == "KA" and r["Counter_Name"] == "GRBM_GUI_ACTIVE": calls[n] += 1
for n in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", 0)):
    c = acc[n]; k = max(calls[n], 1)
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8 / k            # cycles per launch (the counter is summed over the 8 XCDs)
    print(f"{n}\n   launches {k}  cycles/launch {cyc:.0f} ({cyc / 2.4e6:.3f} ms at 2.4 GHz)  MFMA busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / k / max(cyc * 1024, 1):.3f}")
    print("   per launch: " + "  ".join(f"{m}={v / k:.3g}" for m, v in sorted(c.items()) if m != "GRBM_GUI_ACTIVE"))
PY
