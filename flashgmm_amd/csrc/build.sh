#!/usr/bin/env bash
# Build libflashgmm_amd.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
#   -ffp-contract=off                         : the only fused ops are the explicit FMAs of fgmm_math.h
#   -fhip-fp32-correctly-rounded-divide-sqrt  : IEEE '/' and sqrt on the device (bit-exact CDFs)
#   -march=x86-64-v3                          : host rANS code may use AVX2/BMI2, stays portable across hosts
#   -Xarch_device -fno-slp-vectorize          : packed fp32 ops (v_pk_fma_f32 ...) issue at half rate on gfx950 and cost
#                                               pairing moves: measured 1-6 % (symtab) / 12 % (cdftab count) slower with them
# Sources: the kernels (*.hip), the device layer over HIP (fgmm_device_hip.cpp), and the host side, which sees the device only
# through fgmm_device.h (fgmm_capi / fgmm_encode / fgmm_decode / fgmm_decode_gpu / fgmm_rans: plain C++, also built without a GPU
# toolchain against tests/fake/fake_device.cpp by scripts/tsan_host.sh).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${OUT:-../libflashgmm_amd.so}
COMMON="-O3 -fPIC -std=c++17 -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -fno-fast-math"
$HIPCC $COMMON --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -Xarch_device -fno-slp-vectorize \
    -march=x86-64-v3 -shared -o $OUT fgmm_kernels.hip fgmm_tab.hip fgmm_head.hip fgmm_head16.hip fgmm_device_hip.cpp fgmm_rans.cpp fgmm_capi.cpp fgmm_encode.cpp \
    fgmm_decode.cpp fgmm_decode_gpu.cpp -lpthread "$@"
echo "built $(realpath $OUT)"
# The compiled Python boundary over the C ABI (fgmm_pybind.cpp -> flashgmm_amd/_native.*.so): plain C++ against the Python and pybind11
# headers, linked to the library above through an $ORIGIN rpath.  Skipped (the ctypes binding, flashgmm_amd/_lib.py, binds the same
# ABI) when OUT names another library (A/B builds) or the headers are not there.
if [ "$OUT" = "../libflashgmm_amd.so" ] && PYINC=$(python3-config --includes 2>/dev/null) && PB=$(python3 -c 'import pybind11; print(pybind11.get_include())' 2>/dev/null); then
  EXT=$(python3-config --extension-suffix)
  ${CXX:-g++} -O2 -fPIC -shared -std=c++17 -Wall -fvisibility=hidden $PYINC -I"$PB" fgmm_pybind.cpp -o ../_native$EXT -L.. -lflashgmm_amd -Wl,-rpath,'$ORIGIN'
  echo "built $(realpath ../_native$EXT)"
else
  echo "(flashgmm_amd/_native not built: pybind11 / Python headers missing or OUT set - the ctypes binding is used)"
fi
