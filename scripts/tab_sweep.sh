#!/usr/bin/env bash
# Dev aid (GPU box): decode-side table kernel time against the LDS budget per block, then a rocprofv3 kernel table
set -uo pipefail
out=gpurun_out/${1:-tabsweep}
mkdir -p "$out"
for cap in 4096 8192 16384 32768 65536; do
  FGMM_TAB_CAP_E=$cap python bench.py --steps 8 --warmup 2 --no-cpu-baseline > "$out/bench_cap$cap.json" 2> "$out/bench_cap$cap.err"
  python3 -c 'import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2],d["value"],d["ms_per_step"],d["kernels_ms"])' "$out/bench_cap$cap.json" $cap
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > "$out/bench_prof.json" 2> "$out/prof.err"
f=$(ls $out/prof/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats.csv"; head -12 "$out/kernel_stats.csv"
