#!/usr/bin/env bash
# Dev aid (GPU box): end-to-end step time against the decode tail window settings
set -uo pipefail
out=gpurun_out/${1:-tailsweep}
mkdir -p "$out"
for cfg in "8 4" "12 4" "16 4" "16 6" "16 8" "12 6" "8 8"; do
  set -- $cfg
  FGMM_TAIL_ITEMS=$1 FGMM_TAIL_PIECES=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$out/b_$1_$2.json" 2> "$out/b_$1_$2.err"
  python3 -c 'import json,sys;d=json.load(open(sys.argv[1]));print("tail",sys.argv[2],sys.argv[3],d["value"],d["ms_per_step"],d["kernels_ms"]["tab_kernels_all_launches"])' "$out/b_$1_$2.json" $1 $2
done
