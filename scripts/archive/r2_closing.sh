#!/bin/bash
# round-2 closing pass (after the lane kernel): every BASELINE configuration on one device, the default bench line with the bounded CPU
# baseline, the self-launched 2-rank gloo rehearsal, the simulated 8-rank shard
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2z; mkdir -p $O
bash scripts/bench_configs.sh 2>&1 | tee $O/bench_configs.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
python bench.py --gpus 2 --backend gloo --nobs 2000000 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"; tail -c 300 $O/bench_gloo2.json; echo
python bench.py --sim-world 8 --force-dist --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sim-world 8: ms/step %.4f kernel ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
