#!/usr/bin/env python3
"""A/B variant of the library that differs in ONE compilation unit -- the plain unit of elbo_mlp.hip, with --lane the units of
elbo_lane.hip, with --unit=STEM the unit of that object stem in careless_amd/build.py (elbo_mlp_imgl, elbo_narrow, ...):

    python scripts/build_variant.py [--lane | --unit=STEM] [--src=FILE] NAME -DFLAG=... [-DFLAG2=...]   ->   careless_amd/lib/exp_NAME.so

--src=FILE compiles the varied unit from FILE instead of the tree's source (an older revision: `git show HEAD:path > /tmp/old.hip`).

The other units come from a cache of objects under /tmp/t/base_<source hash> (built once per state of the sources, in parallel).
Prints the register / spill figures of the bench instance (5 x 64, d <= 32)."""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from careless_amd import build as B      # noqa: E402

argv = sys.argv[1:]
lane = "--lane" in argv
unit = next((a.split("=", 1)[1] for a in argv if a.startswith("--unit=")), None)
alt_src = next((a.split("=", 1)[1] for a in argv if a.startswith("--src=")), None)
argv = [a for a in argv if a != "--lane" and not a.startswith("--unit=") and not a.startswith("--src=")]
name, flags = argv[0], argv[1:]
hipcc = B._hipcc()
base = f"/tmp/t/base_{B.source_hash()}"
os.makedirs(base, exist_ok=True)
common = [hipcc, f"--offload-arch={B.ARCH}", "-O3", "-fPIC", "-std=c++17"]
procs, objs = [], []
varied = [u for u in B.UNITS if (u[1] == unit if unit else (u[1].startswith("elbo_lane") if lane else u[1] == "elbo_mlp"))]
for src, stem, fl in B.UNITS:
    if (src, stem, fl) in varied:
        continue
    o = os.path.join(base, stem + ".o")
    objs.append(o)
    if not os.path.exists(o):
        procs.append(subprocess.Popen(common + fl + ["-c", os.path.join(B.CSRC, src), "-o", o]))
exps, runs = [], []
for src, stem, fl in varied:
    exp = f"/tmp/t/exp_{name}_{stem}.o"
    exps.append(exp)
    runs.append((stem, subprocess.Popen(common + fl + flags + ["-I", B.CSRC, "-c", alt_src or os.path.join(B.CSRC, src), "-o", exp, "-Rpass-analysis=kernel-resource-usage"],
                                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
want = "_Z16elbo_lane_kernelILi10E" if lane else "_Z15elbo_mlp_kernelILi64ELi32ELi5ELi0E"
for stem, pr in runs:
    _, err = pr.communicate()
    if pr.returncode != 0:
        print(err[-3000:])
        sys.exit(1)
    for b in err.split("Function Name: "):
        if b.startswith(want):
            keep = [ln for ln in b.splitlines() if re.search(r"VGPRs:|AGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize", ln)]
            print(b.split()[0][:48], " | ".join(re.sub(r".*remark: *", "", k).replace("[-Rpass-analysis=kernel-resource-usage]", "").strip() for k in keep))
for p in procs:
    if p.wait() != 0:
        sys.exit("base unit failed")
out = os.path.join(B.LIBDIR, f"exp_{name}.so")
subprocess.check_call([hipcc, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out] + objs + exps)
print("built", out)
