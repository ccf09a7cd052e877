#!/usr/bin/env bash
# GPU box: round 4's evidence.  Everything under gpurun_out/$1 (copied to profiles/r04_* afterwards).
#   the driver's bench command (with the modes / elic4k legs) + three repeats, the ELIC-4K line, one line per approximation mode,
#   the two-rank rehearsal on one device, rocprofv3 kernel tables of the same commands, PMC traffic passes (kodak24 f32, elic4k
#   f16), the checkpointed step under the profiler
set -uo pipefail
out=gpurun_out/${1:-r04prof}
mkdir -p "$out"
note() { echo "[profile_r04] $*"; }
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_unprofiled.json" 2> "$out/bench_unprofiled.err" || { tail -5 "$out/bench_unprofiled.err"; exit 1; }
note "bench done"
for r in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sublegs >> "$out/bench_repeats.jsonl" 2>> "$out/bench_repeats.err"; done
note "repeats done"
python bench.py --workload elic4k --steps 5 --warmup 2 > "$out/bench_elic4k.json" 2> "$out/bench_elic4k.err" || { tail -5 "$out/bench_elic4k.err"; exit 1; }
note "elic done"
for m in polya as logistic; do python bench.py --mode $m --steps 20 --warmup 5 --no-cpu-baseline --no-sublegs > "$out/sweep_kodak24_$m.json" 2>> "$out/sweep.err"; done
note "mode sweep done"
FGMM_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --images 12 --steps 10 --warmup 3 --no-cpu-baseline > "$out/n2_one_device.json" 2> "$out/n2_one_device.err" || tail -5 "$out/n2_one_device.err"
note "two-rank rehearsal done"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$out/bench_under_rocprof.json" 2> "$out/prof.err"
f=$(ls $out/prof/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats.csv"
note "rocprof kodak done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_elic" -- python3 bench.py --workload elic4k --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$out/bench_elic4k_under_rocprof.json" 2> "$out/prof_elic.err"
f=$(ls $out/prof_elic/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats_elic4k.csv"
note "rocprof elic done"
timeout -k 10 300 bash scripts/collect_pmc.sh "$out/pmc" polya kodak24 4 > "$out/pmc.log" 2>&1
cp "$out/pmc/pmc_symtab.json" "$out/pmc_symtab.json"
note "pmc kodak done"
timeout -k 10 300 bash scripts/collect_pmc.sh "$out/pmc_elic" polya elic4k 2 > "$out/pmc_elic.log" 2>&1
cp "$out/pmc_elic/pmc_symtab.json" "$out/pmc_symtab_elic4k.json"
note "pmc elic done"
rm -rf "$out/prof" "$out/prof_elic" "$out/pmc/pmc_fetch" "$out/pmc/pmc_write" "$out/pmc_elic/pmc_fetch" "$out/pmc_elic/pmc_write"
CKPT=1024 ROUNDS=2 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_ck" -- python3 scripts/ab_options.py codec > "$out/ck_codec_1024.txt" 2> "$out/prof_ck.err"
f=$(ls $out/prof_ck/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats_checkpointed.csv"; rm -rf "$out/prof_ck"
note "checkpointed profile done"
head -6 "$out/kernel_stats.csv" | cut -c1-160; cat "$out/pmc_symtab.json" "$out/pmc_symtab_elic4k.json"
python3 -c 'import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]);print(f.split("/")[-1], d["value"],d["ms_per_step"],d["step_ms"]["median"],d["roofline"]["frac"],d["roofline_decode"]["ms_per_step"],d.get("upper_bound",{}).get("value"),d.get("checkpointed",{}).get("value"))' "$out/bench_unprofiled.json" "$out/bench_elic4k.json" "$out/sweep_kodak24_polya.json" "$out/sweep_kodak24_as.json" "$out/sweep_kodak24_logistic.json" "$out/n2_one_device.json"
