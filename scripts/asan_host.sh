#!/usr/bin/env bash
# The real library's host side under AddressSanitizer + UBSan, CPU tests only: the table path, the host rANS building blocks, the
# loader (GPU ASan is not available on this pool).  The CONCURRENT pipeline runs under TSan / ASan on the fake device: scripts/tsan_host.sh.  Builds a throw-away variant of the library and points the ctypes loader at it.
set -euo pipefail
cd "$(dirname "$0")/.."
mkdir -p scripts/bin
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
(cd flashgmm_amd/csrc && /opt/rocm/bin/hipcc -O1 -g -fPIC -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 \
    -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -march=x86-64-v3 -Xarch_host -fsanitize=address \
    -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -shared -shared-libsan \
    -o ../../scripts/bin/libfgmm_asan.so fgmm_kernels.hip fgmm_tab.hip fgmm_device_hip.cpp fgmm_rans.cpp fgmm_capi.cpp fgmm_encode.cpp fgmm_decode.cpp fgmm_decode_gpu.cpp -lpthread)
FGMM_LIB=$PWD/scripts/bin/libfgmm_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests -q -m "not gpu" -k "not dropin and not parallel" 2>&1 | tee scripts/bin/asan.log | tail -3
if grep -q "runtime error\|AddressSanitizer" scripts/bin/asan.log; then echo "SANITIZER REPORTS FOUND"; exit 1; fi
echo "sanitizers: clean"
rm -f scripts/bin/libfgmm_asan.so
