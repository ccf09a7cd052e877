#!/usr/bin/env bash
# Dev aid (GPU box): what bounds segdec_kernel?  SQ counters of the checkpointed decode (scripts/ab_options.py, CKPT=stride,
# gpu_decode=1), one --pmc pass per group, plus one kernel-trace pass for its duration.   bash scripts/pmc_segdec.sh [stride] [outdir]
set -uo pipefail
stride=${1:-1024}
out=gpurun_out/${2:-pmcsegdec}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CKPT=$stride ROUNDS=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 scripts/ab_options.py all gpu_decode=1 > "$out/trace.log" 2> "$out/trace.err" || tail -3 "$out/trace.err"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/p$i" -- python3 scripts/ab_options.py all gpu_decode=1 > "$out/p$i.log" 2> "$out/p$i.err" || tail -3 "$out/p$i.err"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "segdec_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
for f in glob.glob(f"{out}/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "segdec" in r["Name"] or "symtab" in r["Name"]:
            print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
