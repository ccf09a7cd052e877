#!/usr/bin/env bash
# Dev aid (GPU box): scripts/l3_ab.sh for the ELIC-4K workload (eleven calls per step, lists of tensors: another glue path)
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${1:-3}); do
  for s in 0 1; do
    FGMM_BENCH_L3=$s python bench.py --workload elic4k --steps 4 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/l3e_${s}_${rep}.json 2> /dev/null || exit 1
    python3 - "$s" gpurun_out/l3e_${s}_${rep}.json <<'P'
import json, sys
d = json.load(open(sys.argv[2])); p = d["step_ms"]["phases_ms"]
print(f"own L3 {sys.argv[1]}: {d['value']:7.1f} Mpixels/s  step median {d['step_ms']['median']:.2f}  between_calls {p['between_calls']:.2f}  bus {sum(v for k, v in p.items() if k.endswith('.bus')):.1f}  "
      f"tails {sum(v for k, v in p.items() if k.endswith('.host_tail')):.1f}  heads {sum(v for k, v in p.items() if k.endswith('.head')):.2f}  cpu_ms {d['step_ms']['cpu_ms'][0]}")
P
  done
done
