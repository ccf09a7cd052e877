#!/usr/bin/env bash
# Dev aid (GPU box): bench.py's headline with and without settle_calling_thread(), taking turns in one box.   scripts/settle_ab.sh [rounds]
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${1:-4}); do
  for s in 0 1; do
    FGMM_BENCH_SETTLE=$s python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/settle_${s}_${rep}.json 2> /dev/null || exit 1
    python3 - "$s" gpurun_out/settle_${s}_${rep}.json <<'P'
import json, sys
d = json.load(open(sys.argv[2])); p = d["step_ms"]["phases_ms"]; c = d["step_ms"].get("calling_thread")
print(f"settle {sys.argv[1]}: {d['value']:7.1f} Mpixels/s  step median {d['step_ms']['median']:.3f} p90 {d['step_ms']['p90']:.3f}  between_calls {p['between_calls']:.3f}  "
      f"bus {p['call1_decode.bus']:.3f} {p['call2_decode.bus']:.3f}  " + (f"cpu {c['cpu_before']} -> {c['cpu']}  work {c['work_ms_before']} -> {c['work_ms']}" if c else ""))
P
  done
done
