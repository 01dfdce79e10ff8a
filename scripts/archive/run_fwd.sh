for r in 1 2; do for v in "$@"; do CARELESS_HIP_LIB=$PWD/careless_amd/lib/exp_$v.so python scripts/fwd_probe.py 2>/dev/null | tail -1 | sed "s#$PWD/careless_amd/lib/##"; done; done
