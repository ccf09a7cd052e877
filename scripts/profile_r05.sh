#!/usr/bin/env bash
# GPU box: round 5's evidence, part 1.  Everything under gpurun_out/$1 (copied to profiles/r05_* afterwards).
#   the driver's bench command (with its sub-legs and step_diag) + three repeats, the ELIC-4K line, the two-rank rehearsal on one
#   device, rocprofv3 kernel tables of the same commands
set -uo pipefail
out=gpurun_out/${1:-r05prof}
mkdir -p "$out"
note() { echo "[profile_r05] $*"; }
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_unprofiled.json" 2> "$out/bench_unprofiled.err" || { tail -5 "$out/bench_unprofiled.err"; exit 1; }
note "bench done"
for r in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sublegs >> "$out/bench_repeats.jsonl" 2>> "$out/bench_repeats.err"; done
note "repeats done"
python bench.py --workload elic4k --steps 5 --warmup 2 --diag-steps 0 > "$out/bench_elic4k.json" 2> "$out/bench_elic4k.err" || { tail -5 "$out/bench_elic4k.err"; exit 1; }
note "elic done"
FGMM_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --images 12 --steps 10 --warmup 3 --no-cpu-baseline > "$out/n2_one_device.json" 2> "$out/n2_one_device.err" || tail -5 "$out/n2_one_device.err"
note "two-rank rehearsal done"
FGMM_BENCH_PG=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sublegs --diag-steps 0 > "$out/bench_n1_rccl_group.json" 2> "$out/bench_n1_rccl_group.err" || tail -5 "$out/bench_n1_rccl_group.err"
note "N=1 with an RCCL group done"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$out/bench_under_rocprof.json" 2> "$out/prof.err"
f=$(ls $out/prof/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats.csv"
note "rocprof kodak done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_elic" -- python3 bench.py --workload elic4k --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$out/bench_elic4k_under_rocprof.json" 2> "$out/prof_elic.err"
f=$(ls $out/prof_elic/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats_elic4k.csv"
note "rocprof elic done"
rm -rf "$out/prof" "$out/prof_elic"
head -6 "$out/kernel_stats.csv" | cut -c1-160
python3 -c 'import json,sys
for f in sys.argv[1:]:
    for ln in open(f).read().strip().splitlines():
        d=json.loads(ln);print(f.split("/")[-1], d["value"],d["ms_per_step"],d["step_ms"]["median"],d["step_ms"]["p90"],d["roofline"]["frac"],d.get("upper_bound",{}).get("value"),d.get("checkpointed",{}).get("value"))' "$out/bench_unprofiled.json" "$out/bench_repeats.jsonl" "$out/bench_elic4k.json" "$out/n2_one_device.json" "$out/bench_n1_rccl_group.json"
