#!/usr/bin/env bash
# Dev aid (GPU box): end-to-end step time against the decode pipeline settings: "pieces dec_first dec_group"
set -uo pipefail
out=gpurun_out/${1:-piecesweep}
mkdir -p "$out"
IFS=";" read -ra CFGS <<< "${SWEEP_CFGS:-4 2 0;2 2 0;3 2 0;6 2 0;4 1 0;4 4 0}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  FGMM_PIECES=$1 FGMM_DEC_FIRST=$2 FGMM_DEC_GROUP=$3 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$out/b_$1_$2_$3.json" 2> "$out/b_$1_$2_$3.err"
  python3 -c 'import json,sys;d=json.load(open(sys.argv[1]));print("pieces/first/group",sys.argv[2:],d["value"],d["ms_per_step"])' "$out/b_$1_$2_$3.json" $1 $2 $3
done
