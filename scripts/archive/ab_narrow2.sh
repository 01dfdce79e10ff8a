#!/bin/bash
# bash scripts/ab_narrow2.sh variant...  : narrow-kernel parity cases (first variant only), then A/B on the CLI-default workload
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
first=$1
CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$first.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --no-header -k "cli_default or mlp9x7 or mlp7x12 or mlp5x13 or trajectory or three_obs or rank_shards" 2>&1 | tail -3
bash scripts/ab_narrow.sh mono_10M_cli_default_20x10_S1 "$@"
