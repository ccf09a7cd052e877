#!/usr/bin/env bash
# Dev aid (any box; CPU only): ns/symbol of the host decoder on one Kodak half, by library build and Elias-Fano threshold
#   lib_hb_{old,new}{14,49}.so: without / with the straight-line search of long (8-low-bit) rows, EF rows from 14 / 49 entries on
set -uo pipefail
out=gpurun_out/${1:-hostbench}
mkdir -p "$out"
lscpu | grep -E "Model name|MHz" | head -3 | tee "$out/host_decoder.txt"
for rep in 1 2; do
  for v in old14 new14 old49 new49; do
    echo "== $v" | tee -a "$out/host_decoder.txt"
    EF_MIN=${v#???} FGMM_LIB=$PWD/scripts/bin/lib_hb_$v.so taskset -c 3 python scripts/host_bench.py 2>&1 | tail -2 | tee -a "$out/host_decoder.txt"
  done
done
