#!/bin/bash
# After scripts/r6_final.sh (one gpurun call) has merged its files into gpurun_out/: copy the round's evidence into profiles/ (tracked), rebuild
# profiles/traffic.json and profiles/INDEX.md, print DESIGN section 6's table.   bash scripts/r6_store.sh
set -e
bash scripts/store_profiles.sh r6 | tail -1

cp gpurun_out/r6/gpu_suite.txt profiles/r6_gpu_suite.txt

cp gpurun_out/r6_rehearsal_gloo.txt profiles/r6_rehearsal_gloo.txt
cp gpurun_out/r6/envelope.txt profiles/r6_envelope.txt
cp gpurun_out/r6/envelope_before.txt profiles/r6_envelope_before.txt
for f in gpurun_out/r6/kernel_stats_frozen_*.csv gpurun_out/r6/pmc_frozen_*.txt; do cp $f profiles/r6_$(basename $f); done
python3 - <<'PY'
import json
out = ["The step of a training whose scaling model is frozen (--freeze-scales; the half-dataset trainings of --merge-half-datasets), one MI355X, final sources",
       "(scripts/r6_final.sh -> scripts/frozen_bench.py: 50 event-timed steps; data term = what replaces the fused kernel, timed alone; algorithmic bytes",
       " 28 N + 8 R S, harmonic groups + (12 + 8 S) N; cl_slot_rows = round 5's path: slot kernels on plain rows, a float atomic per (row, sample)):", ""]
for l in open('gpurun_out/r6/frozen_step.jsonl'):
    r = json.loads(l)
    out.append("%-36s %-15s step %.3f ms   data term %.3f ms   algorithmic %.3f GB   %.2f TB/s = %.3f of 8 TB/s" % (
        r['workload'], r['path'], r['ms_per_step'], r['data_term_ms'], r['algorithmic_GB'], r['data_term_TBps'], r['frac_of_8TBps']))
open('profiles/r6_frozen_step.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out[4:]))
PY
python3 scripts/design_table.py r6
