#!/usr/bin/env bash
# the shader clock a kernel actually runs at: SQ_BUSY_CYCLES (per shader engine, 32 of them) over the kernel's duration, for the bare
# f32 MFMA loop (scripts/mfma_f32_rate.hip) and for the parameter-head kernel (scripts/head_bench.py)
set -uo pipefail
out=gpurun_out/${1:-clock}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
hipcc -O3 --offload-arch=gfx950 scripts/mfma_f32_rate.hip -o /tmp/mfma_rate 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$out/mfma" -- /tmp/mfma_rate > "$out/mfma.txt" 2> "$out/mfma.err" || tail -3 "$out/mfma.err"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$out/head" -- python3 scripts/head_bench.py 24 4 > "$out/head.json" 2> "$out/head.err" || tail -3 "$out/head.err"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("mfma", "head"):
    dur = {}
    for f in glob.glob(f"{out}/{tag}/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[(r["Dispatch_Id"])] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X", ""))
    rows = collections.defaultdict(dict)
    for f in glob.glob(f"{out}/{tag}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    for did, c in sorted(rows.items(), key=lambda kv: int(kv[0])):
        name, ns, grid = dur.get(did, ("?", 0, ""))
        if ns < 200_000 or not ("head_kernel" in name or name.startswith("void k<") or "k<" in name):
            continue
        busy, gui, mf = c.get("SQ_BUSY_CYCLES", 0), c.get("GRBM_GUI_ACTIVE", 0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        print(f"{tag} {name[:48]:48s} grid {grid:>8s} {ns / 1e3:9.1f} us  SQ_BUSY/32 per us {busy / 32 / (ns / 1e3):7.1f}  GRBM_GUI_ACTIVE per us {gui / (ns / 1e3):8.1f}  MFMA busy per SIMD-us {mf / 1024 / (ns / 1e3):7.1f}")
PY
