#!/usr/bin/env bash
# what the chip's clock and power do under the parameter-head kernel (f32 MFMA + LDS + L2 traffic) against the bare MFMA loop of
# scripts/mfma_f32_rate.hip: rocm-smi sampled while each runs
set -u
sample() { for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr -s ' ' | tr '\n' '|'; echo; sleep 0.4; done; }
hipcc -O3 --offload-arch=gfx950 scripts/mfma_f32_rate.hip -o /tmp/mfma_rate 2>/dev/null
echo "== idle"; sample | head -2
echo "== bare MFMA loop"
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do /tmp/mfma_rate > /dev/null; done ) &
sleep 1.5; sample; wait
echo "== head kernel (c_in 2560, 60 reps)"
python scripts/head_bench.py 24 60 2560 > /tmp/hb.json 2>/dev/null &
sleep 9; sample; wait
python -c "import json; d=json.load(open('/tmp/hb.json')); print(d['unfused'], d['fused'])"
