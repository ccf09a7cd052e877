#!/usr/bin/env bash
# Build libflashgmm_amd.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
#   -ffp-contract=off                         : the only fused ops are the explicit FMAs of fgmm_math.h
#   -fhip-fp32-correctly-rounded-divide-sqrt  : IEEE '/' and sqrt on the device (bit-exact CDFs)
#   -march=x86-64-v3                          : host rANS code may use AVX2/BMI2, stays portable across hosts
#   -Xarch_device -fno-slp-vectorize          : packed fp32 ops (v_pk_fma_f32 ...) issue at half rate on gfx950 and cost
#                                               pairing moves: measured 1-6 % (symtab) / 12 % (cdftab count) slower with them
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${OUT:-../libflashgmm_amd.so}
COMMON="-O3 -fPIC -std=c++17 -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -fno-fast-math"
$HIPCC $COMMON --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -Xarch_device -fno-slp-vectorize \
    -march=x86-64-v3 -shared -o $OUT fgmm_kernels.hip fgmm_tab.hip fgmm_rans.cpp fgmm_capi.cpp -lpthread -lhsa-runtime64 "$@"
echo "built $(realpath $OUT)"
