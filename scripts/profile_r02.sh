#!/usr/bin/env bash
# GPU box: the round's evidence — bench line, rocprofv3 kernel table of the same command, PMC traffic passes.
set -uo pipefail
out=gpurun_out/${1:-r02prof}
mkdir -p "$out"
python bench.py --steps 20 --warmup 3 > "$out/bench_unprofiled.json" 2> "$out/bench_unprofiled.err"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$out/bench_under_rocprof.json" 2> "$out/prof.err"
f=$(ls $out/prof/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats.csv"
bash scripts/collect_pmc.sh "$out/pmc" polya > "$out/pmc.log" 2>&1
cp "$out/pmc/pmc_symtab.json" "$out/pmc_symtab.json"
head -8 "$out/kernel_stats.csv"; cat "$out/pmc_symtab.json"; python3 -c 'import json,sys;d=json.load(open(sys.argv[1]));print(d["value"],d["ms_per_step"],d["roofline"],d["roofline_decode"])' "$out/bench_unprofiled.json"
