#!/usr/bin/env bash
# Dev aid (GPU box): end-to-end step time against the decode pipeline settings: "tail_items tail_pieces dec_first dec_group"
set -uo pipefail
out=gpurun_out/${1:-tailsweep}
mkdir -p "$out"
IFS=";" read -ra CFGS <<< "${SWEEP_CFGS:-8 4 2 0;16 4 2 0;16 3 1 0;8 4 2 0;16 4 2 0;16 3 1 0;12 4 2 0;16 4 1 0}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  FGMM_TAIL_ITEMS=$1 FGMM_TAIL_PIECES=$2 FGMM_DEC_FIRST=$3 FGMM_DEC_GROUP=$4 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$out/b_$1_$2_$3_$4.json" 2> "$out/b_$1_$2_$3_$4.err"
  python3 -c 'import json,sys;d=json.load(open(sys.argv[1]));print("tail/pieces/first/group",sys.argv[2:],d["value"],d["ms_per_step"])' "$out/b_$1_$2_$3_$4.json" $1 $2 $3 $4
done
