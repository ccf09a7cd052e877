#!/usr/bin/env bash
# GPU box: round 6's profiler evidence for the bench's own kernels.  Everything under gpurun_out/$1 (copied to profiles/r06_* afterwards):
#   rocprofv3 --kernel-trace --stats of the driver's command (kodak24), the symtab kernel's HBM traffic by PMC (FETCH_SIZE / WRITE_SIZE in
#   passes of their own, gfx950-corrected: scripts/collect_pmc.sh)
set -uo pipefail
out=gpurun_out/${1:-r06prof}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
FGMM_BENCH_DETAIL="$out/bench_under_rocprof_detail.json" rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$out/bench_under_rocprof.json" 2> "$out/prof.err"
f=$(ls $out/prof/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats.csv"
rm -rf "$out/prof"
head -8 "$out/kernel_stats.csv" | cut -c1-170
timeout -k 10 300 bash scripts/collect_pmc.sh "$out/pmc" polya kodak24 4 > "$out/pmc.log" 2>&1
cp "$out/pmc/pmc_symtab.json" "$out/pmc_symtab.json"
rm -rf "$out/pmc/pmc_fetch"/*/ "$out/pmc/pmc_write"/*/
cut -c1-700 "$out/pmc_symtab.json"
