#!/bin/bash
# round-2 closing pass: whole -m gpu suite, smoke, the default bench line, every BASELINE configuration, kernel-trace stats of the
# image-layer workload, the self-launched 2-rank gloo rehearsal
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
python -m pytest tests -m gpu -q -x --no-header > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json; echo
bash scripts/bench_configs.sh 2>&1 | tee $O/bench_configs.txt
python bench.py --gpus 2 --backend gloo --nobs 2000000 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"; tail -c 300 $O/bench_gloo2.json; echo
wl=mono_10M_studentt_posenc_4x64_img1_S8
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
f=$(find $O/stats_$wl -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$wl.csv; cut -d, -f1-4 $f | cut -c1-110 | head -4; rm -rf $O/stats_$wl
