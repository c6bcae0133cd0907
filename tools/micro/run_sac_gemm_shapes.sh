#!/bin/bash
# build + run tools/micro/sac_gemm_shapes with every ablation (on the GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I dril.jl_amd/csrc -Wno-unused-value -o /tmp/sac_gemm_shapes tools/micro/sac_gemm_shapes.hip || exit 1
for d in 0 16 32 64 48 112; do echo "== DRIL_GEMM_DBG=$d"; DRIL_DEBUG=1 DRIL_GEMM_DBG=$d /tmp/sac_gemm_shapes; done
