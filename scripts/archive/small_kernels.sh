# per-kernel times of the replicated (non-sharded) part of a step, per experimental library: bash scripts/small_kernels.sh variant...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sk_$v -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --force-dist --sim-world 8 > gpurun_out/sk_$v.log 2>&1
  python3 - $v <<'PY'
import csv,glob,sys
v=sys.argv[1]
for f in glob.glob(f"gpurun_out/sk_{v}/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("tn_", "adam", "reduce_partials", "finalize", "elbo_mlp")):
            print(v, r["Name"][:40], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
  tail -1 gpurun_out/sk_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v step', d['ms_per_step'])"
done
