#!/usr/bin/env bash
# Dev aid (GPU box): is the symtab kernel VALU-bound?  SQ counters of the bench's kernels, one --pmc pass per group.
set -uo pipefail
out=gpurun_out/${1:-pmcvalu}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in kodak24 elic4k; do
  steps=3; [ $wl = elic4k ] && steps=1
  i=0
  # GROUPS=quick: the two groups the issue-roof estimate needs (instructions, transcendental ones, waves)
  if [ "${GROUPS_:-all}" = quick ]; then set -- "$1"; grps=("SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS"); else
  grps=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS"); fi
  for grp in "${grps[@]}"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/${wl}_$i" -- python3 bench.py --steps $steps --warmup 1 --workload $wl --no-cpu-baseline --no-extras --schedule all-at-once ${EXTRA:-} > "$out/${wl}_$i.json" 2> "$out/${wl}_$i.err" || tail -3 "$out/${wl}_$i.err"
  done
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for wl in ("kodak24", "elic4k"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{wl}_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "symtab_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(wl, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
