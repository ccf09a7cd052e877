#!/usr/bin/env bash
# Dev aid (GPU box): does limiting the workgroups of the runtime's blit copy (DEBUG_CLR_LIMIT_BLIT_WG) pay, and does it take
# effect when exported from Python after `import torch` but before the first GPU call?  Alternating runs of the bench.
set -uo pipefail
out=gpurun_out/${1:-blitwg}
mkdir -p "$out"
show='import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2],d["value"],d["ms_per_step"],d["step_ms"]["median"],"ub",d.get("upper_bound",{}).get("value"),d["kernels_ms"])'
for rep in 1 2 3; do
  for wg in 0 16 32; do
    if [ $wg = 0 ]; then unset DEBUG_CLR_LIMIT_BLIT_WG; else export DEBUG_CLR_LIMIT_BLIT_WG=$wg; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$out/b_${wg}_$rep.json" 2> "$out/b_${wg}_$rep.err" || { tail -5 "$out/b_${wg}_$rep.err"; exit 1; }
    python3 -c "$show" "$out/b_${wg}_$rep.json" "env=$wg"
  done
  unset DEBUG_CLR_LIMIT_BLIT_WG
  FGMM_BENCH_BLIT_WG=16 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$out/b_py16_$rep.json" 2> "$out/b_py16_$rep.err" || { tail -5 "$out/b_py16_$rep.err"; exit 1; }
  python3 -c "$show" "$out/b_py16_$rep.json" "python-set=16"
done
