export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}" || exit 1
# usage: bash scripts/pmc_passes_step.sh WORKLOAD   -- PMC passes (separate runs) over the dominant kernel(s) of one bench workload.  Output lines
# "<pass> <counter> <launches per step> <value per STEP>": the counter summed over the dominant kernels' launches of one step (4 steps run).
T=$1
STEPS=4
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmcA_$T -- $B > gpurun_out/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcB_$T -- $B > gpurun_out/pmcB.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcC_$T -- $B > gpurun_out/pmcC.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/pmcD_$T -- $B > gpurun_out/pmcD.log 2>&1
python3 - $T $STEPS <<'PY'
import csv,glob,collections,sys
T, steps = sys.argv[1], int(sys.argv[2])
for d in "ABCD":
    for f in glob.glob(f"gpurun_out/pmc{d}_{T}/*/*counter_collection.csv"):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if any(k in r['Kernel_Name'] for k in ('elbo_mlp', 'elbo_narrow', 'elbo_lane', 'wide_stream', 'wide_sq', 'wide_gemm', 'wide_head', 'peel_')):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(d,k,len(v)//steps,sum(v)/steps)
print('# per STEP: counter summed over the launches of the dominant kernel(s) in one step (column 3 = launches per step).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request: double it (calibrated on the fused kernel\'s own access pattern with scripts/calib_fetch.sh: forward-only launch, 960 MB of metadata -> FETCH_SIZE 469117 KiB; WRITE_SIZE exact, an atomic request is tallied as a 32-B write)')
PY
