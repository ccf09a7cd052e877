#!/usr/bin/env bash
# Dev aid (GPU box): the pool wakes the worker that went idle last (default) against the one idle longest (FGMM_POOL_FIFO=1: what a shared
# condition variable does), taking turns: kodak24 and ELIC-4K.   scripts/pool_order_ab.sh [rounds]
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${1:-4}); do
  for s in fifo lifo; do
    if [ $s = fifo ]; then export FGMM_POOL_FIFO=1; else unset FGMM_POOL_FIFO; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/po_k_${s}_${rep}.json 2> /dev/null || exit 1
    python bench.py --workload elic4k --steps 4 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/po_e_${s}_${rep}.json 2> /dev/null || exit 1
    python3 - "$s" gpurun_out/po_k_${s}_${rep}.json gpurun_out/po_e_${s}_${rep}.json <<'P'
import json, sys
k = json.load(open(sys.argv[2])); e = json.load(open(sys.argv[3])); p = k["step_ms"]["phases_ms"]; q = e["step_ms"]["phases_ms"]
print(f"{sys.argv[1]}: kodak {k['value']:7.1f} (median {k['step_ms']['median']:.3f}, p90 {k['step_ms']['p90']:.3f}, enc bus {p['call0_encode.bus']:.3f}, busy {p['call0_encode.worker_busy']:.1f} {p['call1_decode.worker_busy']:.1f} {p['call2_decode.worker_busy']:.1f}, "
      f"tails {p['call1_decode.host_tail']:.3f} {p['call2_decode.host_tail']:.3f}, between {p['between_calls']:.3f}, cpu {sum(k['step_ms']['cpu_ms']) / len(k['step_ms']['cpu_ms']):.1f})   "
      f"elic {e['value']:6.1f} (median {e['step_ms']['median']:.1f}, tails {sum(v for n, v in q.items() if n.endswith('.host_tail')):.1f}, cpu {e['step_ms']['cpu_ms'][0]})")
P
  done
done
