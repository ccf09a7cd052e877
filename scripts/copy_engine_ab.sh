#!/usr/bin/env bash
# Dev aid (GPU box): where do the table copies run?  (42.8 % of GPU time is __amd_rocclr_copyBuffer.)
#   (b) runtime knobs that choose between shader copies and the SDMA engines, (c) a CU-masked copy stream ("copy_cus").
# Each configuration: 6 rounds x 5 steps of the Kodak batch, codec schedule, in one process.  Output under gpurun_out/$1
set -uo pipefail
out=gpurun_out/${1:-copyab}
mkdir -p "$out"
run() { # name, env...
  local name=$1; shift
  echo "== $name"
  env "$@" ROUNDS=4 python scripts/ab_options.py codec "copy_cus=0" 2>&1 | grep -E "^codec" | tee -a "$out/summary.txt"
}
run default FGMM_X=0
run HSA_ENABLE_SDMA=0 HSA_ENABLE_SDMA=0
run GPU_FORCE_BLIT_COPY_SIZE=0 GPU_FORCE_BLIT_COPY_SIZE=0
run DEBUG_CLR_LIMIT_BLIT_WG=16 DEBUG_CLR_LIMIT_BLIT_WG=16
run DEBUG_CLR_LIMIT_BLIT_WG=64 DEBUG_CLR_LIMIT_BLIT_WG=64
echo "== CU-masked copy stream"
ROUNDS=6 python scripts/ab_options.py codec "copy_cus=0" "copy_cus=8" "copy_cus=16" "copy_cus=32" "copy_cus=64" 2>&1 | grep -E "^codec" | tee -a "$out/summary.txt"
