#!/usr/bin/env bash
# The product's concurrent host pipeline (fgmm_capi / fgmm_encode / fgmm_decode / fgmm_decode_gpu / fgmm_rans) on the FAKE device
# (tests/fake/fake_device.cpp: no GPU), driven by randomized batches (tests/fake/stress_main.cpp), under
#   1. ThreadSanitizer          2. AddressSanitizer + UndefinedBehaviorSanitizer
# GPU sanitizers are not available on this pool: this is the CPU build the round's verdict asked for.   scripts/tsan_host.sh [seconds]
set -euo pipefail
cd "$(dirname "$0")/.."
SECS=${1:-25}
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
command -v "$CXX" >/dev/null || CXX=g++
SRC="flashgmm_amd/csrc/fgmm_capi.cpp flashgmm_amd/csrc/fgmm_encode.cpp flashgmm_amd/csrc/fgmm_decode.cpp flashgmm_amd/csrc/fgmm_decode_gpu.cpp
     flashgmm_amd/csrc/fgmm_rans.cpp tests/fake/fake_device.cpp tests/fake/stress_main.cpp"
mkdir -p scripts/bin
CC_=${CXX/clang++/clang}; [ "$CXX" = g++ ] && CC_=gcc
build() { # name, sanitizer flags
  $CC_ -O1 -g -fPIC $2 -c oracle/fgmm_oracle.c -o scripts/bin/fgmm_oracle_$1.o
  $CXX -O1 -g -std=c++17 -march=x86-64-v3 -ffp-contract=off -fno-fast-math $2 -fno-omit-frame-pointer $SRC scripts/bin/fgmm_oracle_$1.o -lpthread -lm -o scripts/bin/stress_$1
}
build tsan "-fsanitize=thread"
build asan "-fsanitize=address,undefined -fno-sanitize-recover=undefined"
echo "== ThreadSanitizer"
TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1" ./scripts/bin/stress_tsan "$SECS" 1 2>&1 | tee scripts/bin/tsan.log | tail -3
echo "== AddressSanitizer + UBSan"
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 ./scripts/bin/stress_asan "$SECS" 2 2>&1 | tee scripts/bin/asan_fake.log | tail -3
if grep -q "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error\|STRESS FAILURE\|LeakSanitizer" scripts/bin/tsan.log scripts/bin/asan_fake.log; then
  echo "SANITIZER REPORTS FOUND"; exit 1
fi
echo "sanitizers: clean"
