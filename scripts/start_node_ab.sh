#!/usr/bin/env bash
# Dev aid (GPU box): does the NUMA node a bench.py process STARTS on decide how fast its Python runs after bench.py has bound it to the
# GPU's node?  (The interpreter's heap and every imported module are first-touched where the process starts; binding moves the
# threads, not the pages.)  The same command started on the GPU's node and on the other one, twice each, taking turns.
# (Measured in round 5: it does not - two runs started AND kept on the far node: between_calls 1.44 and 0.53 ms; what decides is
# whether the host workers share the calling thread's L3: profiles/r05_stall_diagnosis.md 7.)
cd "$(dirname "$0")/.."
bdf=$(python3 - <<'P'
import sys; sys.path.insert(0, ".")
from flashgmm_amd import parallel as P
try:
    print(P.gpu_pci_address(0))
except LookupError:  # (the KFD topology is not readable on every box: ask the runtime, in this helper process)
    print(P.runtime_pci_address(0))
P
)
gnode=$(cat /sys/bus/pci/devices/$bdf/numa_node)
other=$((1 - gnode))
echo "GPU 0 = $bdf on NUMA node $gnode; nodes: $(ls -d /sys/devices/system/node/node* | wc -l)"
for rep in 1 2; do
  for n in $gnode $other; do
    cpus=$(cat /sys/devices/system/node/node$n/cpulist)
    taskset -c "$cpus" python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/start_node_${n}_${rep}.json 2> /dev/null || exit 1
    python3 - "$n" "$gnode" gpurun_out/start_node_${n}_${rep}.json <<'P'
import json, sys
d = json.load(open(sys.argv[3])); p = d["step_ms"]["phases_ms"]
print(f"started on node {sys.argv[1]} (GPU's node {sys.argv[2]}): {d['value']:7.1f} Mpixels/s  step median {d['step_ms']['median']:.3f}  between_calls {p['between_calls']:.3f}  "
      f"decode bus {p['call1_decode.bus']:.3f} {p['call2_decode.bus']:.3f}  worker busy {p['call1_decode.worker_busy']:.1f} {p['call2_decode.worker_busy']:.1f}  cpu_ms {d['step_ms']['cpu_ms'][0]}")
P
  done
done
