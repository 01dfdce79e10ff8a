#!/bin/bash
# gpurun helper: the vector-unit Dense layer probe (scripts/probe/vecmlp_probe.hip), three chunk sizes
mkdir -p gpurun_out
for c in 2 3 4; do echo "== rows per chunk: $c"; timeout 120 scripts/probe/vecmlp_probe_ch$c; done 2>&1 | tee gpurun_out/vecmlp_probe.txt
