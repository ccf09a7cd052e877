#!/usr/bin/env bash
# Dev aid (GPU box): gpu tests, then bench lines (kodak24 and elic4k); everything under gpurun_out/$1
set -uo pipefail
out=gpurun_out/${1:-check}
mkdir -p "$out"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > "$out/gpu_tests.log" 2>&1
  rc=$?
  tail -4 "$out/gpu_tests.log"
  [ $rc -ne 0 ] && exit $rc
fi
show='import json,sys;d=json.load(open(sys.argv[1]));print(d["value"],d["ms_per_step"],"ub",d.get("upper_bound",{}).get("value"),"lat",d.get("latency_ms"),"1thr",d.get("one_host_thread"));print(" roofline",d["roofline"]["frac"],d["roofline"]["launch_ms"]," decode",{k:d["roofline_decode"][k] for k in ("ms_per_step","mean_edges_per_latent","valu_frac","hbm_frac")}," pcie B/latent",d["pcie"]["decode_table_bytes_per_latent"]);print(" cpu",d.get("cpu_baseline"))'
python bench.py --steps 10 --warmup 3 ${BENCH_ARGS:-} > "$out/bench.json" 2> "$out/bench.err" || { tail -20 "$out/bench.err"; exit 1; }
python3 -c "$show" "$out/bench.json"
if [ "${SKIP_ELIC:-0}" != "1" ]; then
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload elic4k ${BENCH_ARGS:-} > "$out/bench_elic.json" 2> "$out/bench_elic.err" || { tail -20 "$out/bench_elic.err"; exit 1; }
  python3 -c "$show" "$out/bench_elic.json"
fi
