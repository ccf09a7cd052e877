#!/usr/bin/env bash
# GPU box: round 5's evidence, part 2: PMC passes of their own (HBM traffic of the symtab kernel, kodak24 fp32 and elic4k fp16 with the
# 16-byte loads; VALU instruction counts of the fp16 form)
set -uo pipefail
out=gpurun_out/${1:-r05pmc}
mkdir -p "$out"
timeout -k 10 300 bash scripts/collect_pmc.sh "$out/pmc" polya kodak24 4 > "$out/pmc.log" 2>&1
cp "$out/pmc/pmc_symtab.json" "$out/pmc_symtab.json"; echo "[profile_r05b] pmc kodak done"
timeout -k 10 300 bash scripts/collect_pmc.sh "$out/pmc_elic" polya elic4k 2 > "$out/pmc_elic.log" 2>&1
cp "$out/pmc_elic/pmc_symtab.json" "$out/pmc_symtab_elic4k.json"; echo "[profile_r05b] pmc elic done"
rm -rf "$out/pmc/pmc_fetch" "$out/pmc/pmc_write" "$out/pmc_elic/pmc_fetch" "$out/pmc_elic/pmc_write"
GROUPS_=quick bash scripts/pmc_valu.sh "${1:-r05pmc}/valu" > "$out/pmc_valu.txt" 2>&1; tail -2 "$out/pmc_valu.txt" | cut -c1-400
rm -rf "$out/valu"/*/runc
cat "$out/pmc_symtab.json" "$out/pmc_symtab_elic4k.json" | cut -c1-900
