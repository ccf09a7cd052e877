#!/usr/bin/env bash
# GPU box: the round's evidence — the driver's bench command, rocprofv3 kernel table of the same command, PMC traffic passes
# (kodak24 f32 and elic4k f16), the ELIC-4K bench line.  Everything under gpurun_out/$1
set -uo pipefail
out=gpurun_out/${1:-r03prof}
mkdir -p "$out"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_unprofiled.json" 2> "$out/bench_unprofiled.err" || { tail -5 "$out/bench_unprofiled.err"; exit 1; }
for r in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline >> "$out/bench_repeats.jsonl" 2>> "$out/bench_repeats.err"; done
python bench.py --workload elic4k --steps 5 --warmup 2 > "$out/bench_elic4k.json" 2> "$out/bench_elic4k.err" || { tail -5 "$out/bench_elic4k.err"; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$out/bench_under_rocprof.json" 2> "$out/prof.err"
f=$(ls $out/prof/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_elic" -- python3 bench.py --workload elic4k --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$out/bench_elic4k_under_rocprof.json" 2> "$out/prof_elic.err"
f=$(ls $out/prof_elic/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats_elic4k.csv"
bash scripts/collect_pmc.sh "$out/pmc" polya kodak24 4 > "$out/pmc.log" 2>&1
cp "$out/pmc/pmc_symtab.json" "$out/pmc_symtab.json"
bash scripts/collect_pmc.sh "$out/pmc_elic" polya elic4k 2 > "$out/pmc_elic.log" 2>&1
cp "$out/pmc_elic/pmc_symtab.json" "$out/pmc_symtab_elic4k.json"
rm -rf "$out/prof" "$out/prof_elic" "$out/pmc/pmc_fetch" "$out/pmc/pmc_write" "$out/pmc_elic/pmc_fetch" "$out/pmc_elic/pmc_write"
head -6 "$out/kernel_stats.csv" | cut -c1-160; cat "$out/pmc_symtab.json" "$out/pmc_symtab_elic4k.json"
python3 -c 'import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f));print(f.split("/")[-1], d["value"],d["ms_per_step"],d["step_ms"],d["roofline"]["frac"],d["roofline_decode"]["ms_per_step"],d.get("upper_bound",{}).get("value"),d.get("latency_ms"))' "$out/bench_unprofiled.json" "$out/bench_elic4k.json"
# the checkpointed path: the bench's kodak24 step on checkpointed streams under the profiler (segdec_kernel), its SQ counters
CKPT=1024 ROUNDS=2 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_ck" -- python3 scripts/ab_options.py codec > "$out/ck_codec_1024.txt" 2> "$out/prof_ck.err"
f=$(ls $out/prof_ck/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats_checkpointed.csv"; rm -rf "$out/prof_ck"
bash scripts/pmc_segdec.sh 1024 "${1:-r03prof}/pmc_segdec" > "$out/segdec_counters.txt" 2>&1
rm -rf "$out/pmc_segdec"
for st in 256 512 1024 4096; do CKPT=$st ROUNDS=3 python3 scripts/ab_options.py codec 2>&1 | tail -1 | sed "s/^/stride $st: /" >> "$out/ck_strides.txt"; done
cat "$out/ck_strides.txt"; tail -3 "$out/segdec_counters.txt"
