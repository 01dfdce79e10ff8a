#!/bin/bash
# Calibration of FETCH_SIZE / WRITE_SIZE for THIS kernel's access pattern (dword-per-lane buffer loads of feature-major metadata):
# the forward-only launch reads the metadata exactly once (4 * meta_rows(d) bytes per observation) and writes 8 bytes per observation.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calF -- python3 scripts/fwd_probe.py 10000000 21 5 64 > gpurun_out/calF.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/calW -- python3 scripts/fwd_probe.py 10000000 21 5 64 > gpurun_out/calW.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for d in ("calF","calW"):
    for f in glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv"):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'elbo_mlp' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(d,k,len(v),sum(v)/len(v), "KB per forward-only launch; expected metadata bytes 24 rows x 4 B x 10e6 obs = 960000 KB (937500 KiB), stores 80000 KB")
PY
