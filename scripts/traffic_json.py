"""profiles/traffic.json from the PMC pass summaries of scripts/pmc_passes.sh: HBM bytes per launch of the dominant kernel =
2 x FETCH_SIZE (gfx950 tallies a 128-B request at 64 B: MI355X_MICROARCH.md, calibrated on the fused kernel with scripts/calib_fetch.sh)
+ WRITE_SIZE, both reported in KiB, averaged over the launches of the pass.  Records the source hash of the kernels (careless_amd.build.
source_hash) so bench.py can tell a stale figure.   python scripts/traffic_json.py workload=profiles/r3_pmc_<workload>.txt ..."""
import json, os, re, sys
sys.path.insert(0, ".")
from careless_amd.build import source_hash
out_path = "profiles/traffic.json"
out = {}
for arg in sys.argv[1:]:
    wl, path = arg.split("=", 1)
    vals, launches = {}, {}
    for ln in open(path):
        m = re.match(r"^[A-D] (\w+) (\d+) ([0-9.eE+-]+)$", ln.strip())
        if m:
            vals[m.group(1)] = float(m.group(3))
            launches[m.group(1)] = int(m.group(2))
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        print("no FETCH_SIZE / WRITE_SIZE in", path); continue
    out[wl] = {"n_gpus": 1, "hbm_bytes_per_launch": 1024.0 * (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]),
               "fetch_kib_reported": vals["FETCH_SIZE"], "write_kib": vals["WRITE_SIZE"], "atomic_requests": vals.get("TCC_EA0_ATOMIC_sum"),
               "sources": source_hash(), "file": path,
               "launches_per_step": launches.get("FETCH_SIZE"),
               "method": "2 x FETCH_SIZE + WRITE_SIZE summed over the dominant kernel(s)' launches of one step, separate rocprofv3 --pmc passes (scripts/pmc_passes_step.sh)"}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
