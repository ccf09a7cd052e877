#!/usr/bin/env bash
# second-difference rows (option d2_min: rows of at least that many entries) in the bench step: bytes per latent, step time, throttling
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${ROUNDS:-3}); do
  for d in ${D2S:-0 128 96 64 48 32}; do
    echo -n "d2_min $d : "
    FGMM_D2_MIN=$d python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], 'Mpix/s  step median', d['step_ms']['median'], 'min', d['step_ms']['min'], 'max', d['step_ms']['max'], 'throttled', d['step_ms'].get('cpu_throttled',{}).get('nr_throttled'), 'B/latent', d['pcie']['decode_table_bytes_per_latent'], 'tab_ms', d['roofline_decode']['ms_per_step'])"
  done
done
