#!/usr/bin/env bash
# bench.py --host-threads {2,4,6,8,12,16,24,48} on one GPU: the per-GPU rate an 8-rank run would see under an 8-way share of the
# host (a 16-CPU quota for the node leaves a rank ~6 workers), plain and checkpointed schedules
set -u
for t in ${THREADS:-2 4 6 8 12 16 24 48}; do
  FGMM_BENCH_DETAIL=/tmp/ht_detail.json python bench.py --steps 12 --warmup 3 --host-threads $t --no-cpu-baseline --no-sublegs > /tmp/ht_line.json 2>/dev/null
  python - $t <<'PY'
import json, sys, statistics as st
d = json.load(open("/tmp/ht_detail.json")); sm = d["step_ms"]
print(f"host threads {int(sys.argv[1]):3d}: {d['value']:7.1f} Mpixels/s  step median {sm['median']:7.3f} ms  cpu_ms {st.median(sm['cpu_ms']):6.1f}  upper_bound {d['upper_bound']['value']:7.1f}  checkpointed {d['checkpointed']['value']:7.1f}  one image {d['latency_ms']['as_codec']:.2f} ms", flush=True)
PY
done
