#!/usr/bin/env bash
# A/B of an environment switch of the library, taking turns in one box: bench.py's headline region without baseline and sub-legs
#   bash scripts/ab_env.sh NAME A B [rounds=3]     e.g.  bash scripts/ab_env.sh FGMM_DECODE_SMT 1 0
set -u
name=$1; a=$2; b=$3; rounds=${4:-3}
for r in $(seq 1 $rounds); do
  for v in "$a" "$b"; do
    env $name=$v FGMM_BENCH_DETAIL=/tmp/ab_detail.json python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sublegs > /tmp/ab_line.json 2>/dev/null
    python - "$name=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_detail.json"))
sm = d["step_ms"]; ph = sm.get("phases_ms") or {}
import statistics as st
print(f"{sys.argv[1]:24s} value {d['value']:7.1f}  median {sm['median']:6.3f} p90 {sm['p90']:6.3f}  cpu_ms {st.median(sm['cpu_ms']):6.1f}  busy {ph.get('call1_decode.worker_busy')} {ph.get('call2_decode.worker_busy')}  enc busy {ph.get('call0_encode.worker_busy')}  tails {ph.get('call1_decode.host_tail')} {ph.get('call2_decode.host_tail')}  between {ph.get('between_calls')}  ckpt {d.get('checkpointed', {}).get('value')}", flush=True)
PY
  done
done
