#!/usr/bin/env bash
# Dev aid (GPU box): which placement of the process's threads costs what — the same 20 steps of the Kodak batch under five
# bindings, then the driver's bench command.  Everything under gpurun_out/$1
set -uo pipefail
out=gpurun_out/${1:-numa}
mkdir -p "$out"
for b in all thread none other early; do
  python scripts/diag_env.py --bind $b > "$out/diag_$b.txt" 2>&1 || { tail -5 "$out/diag_$b.txt"; exit 1; }
  echo "== $b"; grep -E "^(gpu|bind|bound|codec|all-at-once)" "$out/diag_$b.txt" | cut -c1-330
done
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver.json" 2> "$out/bench_driver.err" || { tail -20 "$out/bench_driver.err"; exit 1; }
python3 -c 'import json,sys;d=json.load(open(sys.argv[1]));print(d["value"],d["ms_per_step"],d.get("step_ms"),"ub",d.get("upper_bound"),d["config"].get("numa"),d["config"].get("host_threads_per_gpu"))' "$out/bench_driver.json"
for t in 16 12; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-threads $t > "$out/bench_t$t.json" 2> "$out/bench_t$t.err" || { tail -20 "$out/bench_t$t.err"; exit 1; }
  python3 -c 'import json,sys;d=json.load(open(sys.argv[1]));print("threads",d["config"]["host_threads_per_gpu"],d["value"],d["ms_per_step"],d.get("step_ms"),"ub",d.get("upper_bound"))' "$out/bench_t$t.json"
done
