#!/usr/bin/env bash
# A/B of the decode-table copy path and the table kernel's LDS budget on the bench step (no CPU baseline, no sub-legs):
#   FGMM_COPY_ENGINE 0 = hipMemcpyAsync (shader copies), 1 = ONE SDMA engine through HSA, 2 = two engines in turn;  FGMM_TAB_CAP_E 12288 / 8192
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for ce in 0 1 2; do
    for cap in 12288; do
      echo -n "copy_engine $ce cap_e $cap : "
      FGMM_COPY_ENGINE=$ce FGMM_TAB_CAP_E=$cap python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sublegs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], 'Mpix/s  step median', d['step_ms']['median'], ' upper', d['upper_bound']['value'], ' tab_ms', d['roofline_decode']['ms_per_step'], ' lat', d['latency_ms']['as_codec'], ' ckpt', d['checkpointed']['value'])"
    done
  done
done
