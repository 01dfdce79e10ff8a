cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# usage: bash scripts/pmc_passes.sh [workload]   (PMC passes over the fused kernel of one bench workload; default = the bench line)
T=${1:-bench}
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline ${1:+--workload $1}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmcA_$T -- $B > gpurun_out/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcB_$T -- $B > gpurun_out/pmcB.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcC_$T -- $B > gpurun_out/pmcC.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/pmcD_$T -- $B > gpurun_out/pmcD.log 2>&1
python3 - $T <<'PY'
import csv,glob,collections,sys
T=sys.argv[1]
for d in "ABCD":
    for f in glob.glob(f"gpurun_out/pmc{d}_{T}/*/*counter_collection.csv"):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if any(k in r['Kernel_Name'] for k in ('elbo_mlp', 'elbo_narrow', 'elbo_lane', 'wide_stream', 'wide_gemm')):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(d,k,len(v),sum(v)/len(v))
print('# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request: double it (calibrated on this kernel\'s own access pattern with scripts/calib_fetch.sh: forward-only launch, 960 MB of metadata -> FETCH_SIZE 469117 KiB; WRITE_SIZE exact)')
PY
