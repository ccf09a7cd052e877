#!/usr/bin/env bash
# tab_kernel alone (scripts/tab_ab.py: 3.1 M Kodak-like latents): look-back placement (FGMM_TAB_PLACE=1) against the cursor (=0)
# over LDS budgets per block (FGMM_TAB_CAP_E: latents per block = cap / 152 edges, rounded down to 16)
cd "$(dirname "$0")/.."
for cap in 16384 12288 10240 8192 6144 4096 2560; do
  for place in 0 1; do
    echo -n "cap_e $cap place $place : "
    FGMM_TAB_CAP_E=$cap FGMM_TAB_PLACE=$place python scripts/tab_ab.py flashgmm_amd/libflashgmm_amd.so
  done
done
