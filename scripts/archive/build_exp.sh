#!/bin/bash
# build an experimental variant of the library: scripts/build_exp.sh NAME -DFLAG...   -> careless_amd/lib/exp_NAME.so
name=$1; shift
cd /root/repo/careless_amd/csrc
O=/tmp/t/exp_$name; mkdir -p $O
for u in "cl_api: " "elbo_mlp:-DCL_IMGL=0" "elbo_mlp_imgl:-DCL_IMGL=1" "elbo_mlp_packed:-DCL_IMGL=2" "elbo_mlp_chain:-DCL_CHAIN=1" "elbo_elem: " "elbo_laue: "; do
  stem=${u%%:*}; fl=${u#*:}; src=${stem%_imgl}; src=${src%_packed}; src=${src%_chain}.hip
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $fl "$@" -c $src -o $O/$stem.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp_$name.so $O/*.o && echo built exp_$name.so
