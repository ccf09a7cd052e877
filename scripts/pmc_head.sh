#!/usr/bin/env bash
# SQ counters of the parameter-head kernel (scripts/head_bench.py under rocprofv3 --pmc, one pass per group): where a wave's cycles go
set -uo pipefail
out=gpurun_out/${1:-pmchead}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/g$i" -- python3 scripts/head_bench.py 24 3 ${C_IN:-640} > "$out/g$i.json" 2> "$out/g$i.err" || tail -3 "$out/g$i.err"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/g*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "head_kernel" in k:
            acc["fused" if "Lb1ELb1EEE" in k.split("head_kernel")[1][:24] or "ELb1ELb1ELb1" in k else "params"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for v, d in acc.items():
    print(v, {k: round(sum(x) / len(x)) for k, x in sorted(d.items())})
PY
