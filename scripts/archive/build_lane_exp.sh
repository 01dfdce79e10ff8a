#!/bin/bash
# variant of the library that differs in elbo_lane.hip only: scripts/build_lane_exp.sh NAME -DFLAG...  -> careless_amd/lib/exp_NAME.so
name=$1; shift
cd /root/repo/careless_amd/csrc
B=/tmp/t/lbase; mkdir -p $B /tmp/t/lexp
if [ ! -f $B/cl_api.o ] || [ -n "$REBASE" ]; then
for u in "cl_api: " "elbo_mlp:-DCL_IMGL=0" "elbo_mlp_imgl:-DCL_IMGL=1" "elbo_mlp_packed:-DCL_IMGL=2" "elbo_mlp_chain:-DCL_CHAIN=1" "elbo_mlp_det:-DCL_DET=1" "wide_gemm: " "elbo_elem: " "elbo_laue: " "elbo_narrow:-fno-slp-vectorize"; do
  stem=${u%%:*}; fl=${u#*:}; src=${stem%_imgl}; src=${src%_packed}; src=${src%_chain}; src=${src%_det}.hip
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $fl -c $src -o $B/$stem.o &
done
wait
fi
for part in 0 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DCL_LANE_PART=$part "$@" -c elbo_lane.hip -o /tmp/t/lexp/$name.$part.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|AGPRs|VGPRs Spill|ScratchSize" | sed 's/.*remark: *//;s/\[-Rpass.*//' | paste - - - - - - | sort | uniq -c &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp_$name.so $B/*.o /tmp/t/lexp/$name.[0-3].o && echo built exp_$name.so
