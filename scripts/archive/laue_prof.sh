cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash scripts/bench_configs.sh 2>&1 | grep "CFG" 
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/laue_prof -- python3 bench.py --workload laue_5M_normal_5x64_S1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/laue_bench.log 2>&1
f=$(find gpurun_out/laue_prof -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | cut -c1-150 | head -14
