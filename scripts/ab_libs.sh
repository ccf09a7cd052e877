#!/usr/bin/env bash
# Dev aid (GPU box): the same scripts/ab_options.py line for several builds of the library (FGMM_LIB), interleaved twice.
#   CKPT=1024 bash scripts/ab_libs.sh "codec gpu_decode=1" ab/lib_a.so ab/lib_b.so ...
args=$1; shift
for rep in 1 2; do for lib in "$@"; do echo -n "$(basename $lib)  "; FGMM_LIB=$PWD/$lib ROUNDS=${ROUNDS:-3} timeout -k 10 120 python3 scripts/ab_options.py $args 2>&1 | tail -1; done; done
