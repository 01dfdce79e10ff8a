#!/bin/bash
# variant of the library that differs in the plain unit of elbo_mlp.hip only: scripts/build_mlp_exp.sh NAME -DFLAG... -> careless_amd/lib/exp_NAME.so
name=$1; shift
cd /root/repo/careless_amd/csrc
B=/tmp/t/base2; mkdir -p $B /tmp/t/mexp
if [ ! -f $B/cl_api.o ] || [ -n "$REBASE" ]; then
for u in "cl_api: " "elbo_mlp_imgl:-DCL_IMGL=1" "elbo_mlp_packed:-DCL_IMGL=2" "elbo_mlp_chain:-DCL_CHAIN=1" "elbo_elem: " "elbo_laue: " "elbo_narrow:-fno-slp-vectorize" "elbo_lane:-mllvm -amdgpu-mfma-vgpr-form=1"; do
  stem=${u%%:*}; fl=${u#*:}; src=${stem%_imgl}; src=${src%_packed}; src=${src%_chain}.hip
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $fl -c $src -o $B/$stem.o &
done
wait
fi
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DCL_IMGL=0 "$@" -c elbo_mlp.hip -o /tmp/t/mexp/$name.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "Li64ELi32ELi5ELi0ELb0ELb0ELb0ELi4E" | grep -E "error|SGPRs Spill|VGPRs Spill|ScratchSize" | sed 's/.*remark: *//;s/\[-Rpass.*//' | paste - - -
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp_$name.so $B/*.o /tmp/t/mexp/$name.o && echo built exp_$name.so
