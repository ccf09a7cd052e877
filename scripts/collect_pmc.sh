#!/usr/bin/env bash
# Collect the HBM-traffic counters of the symtab kernel (roofline.traffic) exactly as MI355X_MICROARCH.md prescribes:
# separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only.  Writes
#   $1/pmc_fetch/..., $1/pmc_write/... (raw rocprofv3 csv) and $1/pmc_symtab.json (per-launch means, corrected).
# gfx950 correction: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> bytes = 2 * FETCH_SIZE KB.
set -euo pipefail
out=${1:-gpurun_out/pmc_r03}
mode=${2:-polya}
workload=${3:-kodak24}
steps=${4:-4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out/pmc_fetch" "$out/pmc_write"
FGMM_BENCH_DETAIL="$out/pmc_fetch/detail.json" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps $steps --warmup 1 --mode "$mode" --workload "$workload" --no-cpu-baseline --no-extras --schedule all-at-once > "$out/pmc_fetch/bench.json" 2> "$out/pmc_fetch/err.txt"
FGMM_BENCH_DETAIL="$out/pmc_write/detail.json" rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps $steps --warmup 1 --mode "$mode" --workload "$workload" --no-cpu-baseline --no-extras --schedule all-at-once > "$out/pmc_write/bench.json" 2> "$out/pmc_write/err.txt"
python3 - "$out" "$mode" "$workload" "$steps" <<'PY'
import csv, glob, json, sys
out, mode, workload, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
def mean_counter(d, name, kernel):
    f = glob.glob(f"{out}/{d}/*/*_counter_collection.csv")[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and kernel in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)
b0 = json.load(open(f"{out}/pmc_write/detail.json"))  # (the whole result: bench.py's stdout is a summary since round 6)
res = {"workload": workload, "mode": mode, "param_dtype": b0["config"]["param_dtype"], "images_per_gpu": b0["config"]["images_per_gpu"],
       "kernel": "symtab_kernel", "algorithmic_bytes_per_launch": b0["roofline"]["bytes_per_launch"]}
def sum_counter(d, name, kernel, steps):
    f = glob.glob(f"{out}/{d}/*/*_counter_collection.csv")[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and kernel in r["Kernel_Name"]]
    return sum(v) / steps, len(v)
for kern, key in (("symtab_kernel", "symtab"),):
    fk, nf = mean_counter("pmc_fetch", "FETCH_SIZE", kern)
    wk, nw = mean_counter("pmc_write", "WRITE_SIZE", kern)
    res[key] = {"FETCH_SIZE_KB_raw": fk, "WRITE_SIZE_KB": wk, "launches": [nf, nw],
                "hbm_bytes_corrected": int(2 * fk * 1024 + wk * 1024)}
res["symtab"]["traffic_over_algorithmic"] = round(res["symtab"]["hbm_bytes_corrected"] / res["algorithmic_bytes_per_launch"], 4)
# the decode-side table kernel: all launches of a step together (steps + 1 steps ran: 1 warm-up + the timed ones)
try:
    fk, nf = sum_counter("pmc_fetch", "FETCH_SIZE", "tab_kernel", steps + 1)
    wk, nw = sum_counter("pmc_write", "WRITE_SIZE", "tab_kernel", steps + 1)
    b = json.load(open(f"{out}/pmc_write/detail.json"))
    alg = b["roofline_decode"]["hbm_bytes_algorithmic"]
    # parameters are read 4 bytes per lane (FETCH_SIZE uncalibrated for that width: reported raw and doubled), rows are written
    # 4 bytes per lane
    res["tab_kernel"] = {"FETCH_SIZE_KB_raw_per_step": fk, "WRITE_SIZE_KB_per_step": wk, "launches": [nf, nw],
                         "hbm_bytes_per_step_raw": int(fk * 1024 + wk * 1024), "hbm_bytes_per_step_fetch_doubled": int(2 * fk * 1024 + wk * 1024),
                         "algorithmic_bytes_per_step": alg,
                         "traffic_over_algorithmic_raw": round((fk * 1024 + wk * 1024) / alg, 3),
                         "traffic_over_algorithmic_fetch_doubled": round((2 * fk * 1024 + wk * 1024) / alg, 3)}
except Exception as e:
    res["tab_kernel"] = {"error": str(e)}
json.dump(res, open(f"{out}/pmc_symtab.json", "w"), indent=1)
print(json.dumps(res))
PY
