#!/usr/bin/env bash
# BASELINE configs[2]: APPROX_MODE sweep on the kodak24 workload + configs[4] (ELIC 4K, fp16 planes), 1 GPU.
set -euo pipefail
out=${1:-gpurun_out/sweep}
mkdir -p "$out"
for m in polya as logistic; do
  python bench.py --steps 15 --warmup 3 --mode $m --no-cpu-baseline > "$out/kodak24_$m.json"
  cat "$out/kodak24_$m.json"
done
python bench.py --workload elic4k --steps 5 --warmup 2 --no-cpu-baseline > "$out/elic4k_f16_polya.json"
cat "$out/elic4k_f16_polya.json"
