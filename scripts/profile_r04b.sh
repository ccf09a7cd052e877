set -uo pipefail
out=gpurun_out/r04prof
mkdir -p $out
for m in polya as logistic; do python bench.py --mode $m --steps 20 --warmup 5 --no-cpu-baseline --no-sublegs > "$out/sweep_kodak24_$m.json" 2>> "$out/sweep.err"; echo "sweep $m done"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 bash scripts/collect_pmc.sh "$out/pmc_elic" polya elic4k 2 > "$out/pmc_elic.log" 2>&1; echo "pmc elic rc=$?"
cp "$out/pmc_elic/pmc_symtab.json" "$out/pmc_symtab_elic4k.json"; cat "$out/pmc_symtab_elic4k.json"
rm -rf "$out/pmc_elic/pmc_fetch" "$out/pmc_elic/pmc_write"
CKPT=1024 ROUNDS=2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_ck" -- python3 scripts/ab_options.py codec > "$out/ck_codec_1024.txt" 2> "$out/prof_ck.err"; echo "ck rc=$?"
f=$(ls $out/prof_ck/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats_checkpointed.csv"; rm -rf "$out/prof_ck"
head -4 "$out/kernel_stats_checkpointed.csv" | cut -c1-150
python3 -c 'import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]);print(f.split("/")[-1], d["value"],d["ms_per_step"],d["step_ms"]["median"],d["roofline"]["frac"],d["roofline"]["launch_ms"],d["roofline_decode"]["ms_per_step"],d.get("upper_bound",{}).get("value"),d.get("checkpointed",{}).get("value"))' "$out/sweep_kodak24_polya.json" "$out/sweep_kodak24_as.json" "$out/sweep_kodak24_logistic.json"
