#!/bin/bash
# usage: res.sh file.hip  -> prints kernel name, VGPR, AGPR, scratch, spills
hipcc --offload-arch=gfx950 -O3 -std=c++17 $2 -c "$1" -o /tmp/t/out.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for line in sys.stdin:
    if 'error' in line: print(line.strip())
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); d={}
    for k in ['VGPRs','AGPRs','ScratchSize \[bytes/lane\]','VGPRs Spill','Occupancy \[waves/SIMD\]']:
        m=re.search(r'remark:\s+'+k+r': (\d+)',line)
        if m: d[k]=m.group(1)
    if 'LDS Size' in line and cur: print(cur[:60], d); cur=None
"
