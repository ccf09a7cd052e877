#!/usr/bin/env bash
# the bench step with MORE host rANS workers than the CPU quota (16 on a GPU box): workers sleep on the copies' events most of a
# call, so a larger pool need not exhaust the quota - does it get throttled (cpu_throttled in the line)?   ROUNDS=6 THREADS="16 32 48 64"
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${ROUNDS:-6}); do
  for t in ${THREADS:-16 32 48 64}; do
    echo -n "host threads $t : "
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --host-threads $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], 'Mpix/s  step median', d['step_ms']['median'], 'min', d['step_ms']['min'], 'max', d['step_ms']['max'], 'throttled', d['step_ms'].get('cpu_throttled',{}).get('nr_throttled'))"
  done
done
