#!/bin/bash
# tools/build_variant.sh NAME "<extra hipcc flags>" [file.hip ...]: a variant of libdril_hip.so that differs in the named translation units only (default: dril_grad_wide.hip),
# as dril.jl_amd/csrc/libdril_NAME.so — for same-box A/Bs (tools/ab_lib.sh; DRIL_HIP_LIBRARY selects it).  STAMPS=1 adds -DDRIL_STAMPS to those units (per-phase stamps of the variant).
set -e
NAME=$1; FLAGS=$2; shift 2 || true; FILES=${@:-dril_grad_wide.hip}
cd "$(dirname "$0")/../dril.jl_amd/csrc"
CXX="-O3 -fno-slp-vectorize -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value -Wno-unused-local-typedef -mllvm -amdgpu-mfma-vgpr-form"
[ -n "$STAMPS" ] && FLAGS="$FLAGS -DDRIL_STAMPS"
OBJS="dril_kernels.o dril_grad_f32.o dril_grad_pair.o dril_grad_wide.o dril_update_small.o dril_api.o dril_sac.o dril_gemm.o dril_generic.o"
[ -n "$STAMPS" ] && OBJS="dril_kernels_stamps.o dril_grad_f32_stamps.o dril_grad_pair_stamps.o dril_grad_wide_stamps.o dril_update_small_stamps.o dril_api_stamps.o dril_sac_stamps.o dril_gemm_stamps.o dril_generic_stamps.o"
for f in $FILES; do
  o=${f%.hip}_$NAME.o
  /opt/rocm/bin/hipcc $CXX $FLAGS -c $f -o $o
  base=${f%.hip}.o; [ -n "$STAMPS" ] && base=${f%.hip}_stamps.o
  OBJS=${OBJS/$base/$o}
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libdril_$NAME.so $OBJS -ldl -Wl,-rpath,/opt/rocm/lib
ls -la libdril_$NAME.so
