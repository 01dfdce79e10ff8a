#!/bin/bash
# variant of the library that differs in elbo_narrow.hip only: scripts/build_narrow_exp.sh NAME -DFLAG...  -> careless_amd/lib/exp_NAME.so
name=$1; shift
cd /root/repo/careless_amd/csrc
B=/tmp/t/base; mkdir -p $B /tmp/t/nexp
if [ ! -f $B/cl_api.o ] || [ -n "$REBASE" ]; then
for u in "cl_api: " "elbo_mlp:-DCL_IMGL=0" "elbo_mlp_imgl:-DCL_IMGL=1" "elbo_mlp_packed:-DCL_IMGL=2" "elbo_mlp_chain:-DCL_CHAIN=1" "elbo_elem: " "elbo_laue: "; do
  stem=${u%%:*}; fl=${u#*:}; src=${stem%_imgl}; src=${src%_packed}; src=${src%_chain}.hip
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $fl -c $src -o $B/$stem.o &
done
wait
fi
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c elbo_narrow.hip -o /tmp/t/nexp/$name.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs Spill|ScratchSize" | sed 's/.*remark: *//;s/\[-Rpass.*//' | paste - - - - - -
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp_$name.so $B/*.o /tmp/t/nexp/$name.o && echo built exp_$name.so
