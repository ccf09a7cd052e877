#!/usr/bin/env bash
# tab_kernel ALONE (one Kodak stage: 24 halves = 3.1 M coded latents in one launch, scripts/tab_ab.py --child), under rocprofv3:
# the kernel-trace statistics and, in a pass of their own, the SQ instruction counters.  The full bench under --kernel-trace inflates
# this kernel 2.5x (572 small launches behind table copies: profiles/r05_bench_kernel_stats.csv is not usable for it); this is the
# evidence bench.py's roofline_decode.alone cites.    bash scripts/profile_tab_alone.sh [outdir under gpurun_out]
set -uo pipefail
out=gpurun_out/${1:-tabalone}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 scripts/tab_ab.py --child > "$out/unprofiled.txt" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 scripts/tab_ab.py --child > "$out/trace.txt" 2> "$out/trace.err" || tail -3 "$out/trace.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc" -- python3 scripts/tab_ab.py --child > "$out/pmc.txt" 2> "$out/pmc.err" || tail -3 "$out/pmc.err"
python3 - "$out" <<'PY'
import csv, glob, json, re, sys, collections
out = sys.argv[1]
res = {"what": "tab_kernel alone: one Kodak stage (24 halves of [1,192,32,24], coded latents only) in ONE launch through fgmm_build_tab_hip, Polya, sigma clamped",
       "script": "scripts/profile_tab_alone.sh"}
m = re.search(r"n (\d+) max_bs (\d+) tl (\d+):\s+([\d.]+) ms.*edges/latent\s+([\d.]+)\s+rows\s+([\d.]+) B/latent", open(f"{out}/unprofiled.txt").read())
n, ms, epl = int(m.group(1)), float(m.group(4)), float(m.group(5))
res.update({"latents_per_launch": n, "max_bs": int(m.group(2)), "tl": int(m.group(3)), "unprofiled_ms_per_launch_hip_events": ms, "edges_per_latent": epl,
            "row_bytes_per_latent": float(m.group(6))})
for f in glob.glob(f"{out}/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "tab_kernel" in r["Name"]:
            res["kernel_trace"] = {"name": r["Name"][:60], "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
            open(f"{out}/tab_kernel_alone_stats.csv", "w").write(open(f).read())
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "tab_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
pmc = {k: sum(v) / len(v) for k, v in acc.items()}
res["pmc_per_launch"] = {k: round(v) for k, v in sorted(pmc.items())}
if "SQ_INSTS_VALU" in pmc and "kernel_trace" in res:
    edges = n * epl
    per_edge = pmc["SQ_INSTS_VALU"] / edges * 64  # wave-level instructions x 64 lanes / edges = lane-instructions per edge
    trans = pmc.get("SQ_INSTS_VALU_TRANS", 0.0)
    # issue cost of the mix (scripts/valu_peak.hip): plain / packed ~4.5-5.2 cycles per wave instruction, transcendental 9
    cyc = (pmc["SQ_INSTS_VALU"] - trans) * 4.8 + trans * 9.0
    t_issue_ms = cyc / (256 * 4 * 2.4e9) * 1e3
    t = res["kernel_trace"]["average_ns"] / 1e6
    res["valu"] = {"valu_wave_insts_per_latent": round(pmc["SQ_INSTS_VALU"] / n, 3), "valu_lane_insts_per_edge": round(per_edge, 1),
                   "trans_share": round(trans / pmc["SQ_INSTS_VALU"], 4), "issue_ms_at_peak": round(t_issue_ms, 4), "launch_ms": round(t, 4),
                   "valu_frac": round(t_issue_ms / t, 4),
                   "note": "SQ_INSTS_VALU x the mix's issue cost (4.8 cycles per plain / packed wave instruction, 9 per transcendental) / (1024 SIMDs x 2.4 GHz) / the launch's duration"}
json.dump(res, open(f"{out}/tab_kernel_alone.json", "w"), indent=1)
print(json.dumps(res))
PY
