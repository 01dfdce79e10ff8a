cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tiny_stats -- python3 scripts/host_overhead.py > gpurun_out/tiny.log 2>&1
f=$(find gpurun_out/tiny_stats -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    c=int(r["Calls"]); a=float(r["AverageNs"]); 
    if c>=1000: print(f'{r["Name"][:60]:60s} calls {c:6d} avg {a/1e3:7.2f} us'); tot+=a*(c/1050)
print("sum of per-step kernel time ~", tot/1e3, "us")
PY
tail -1 gpurun_out/tiny.log
