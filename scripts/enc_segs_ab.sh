#!/usr/bin/env bash
# encode tables landing tail first in segments (option enc_segs) against whole tables: the encode call and the bench step
cd "$(dirname "$0")/.."
for v in 0 1; do echo "== enc_segs=$v"; timeout -k 10 100 python scripts/trace_encode.py enc_segs=$v 2>&1 | grep "encode\] [a-z]\|python total" | tail -5; done
ROUNDS=${ROUNDS:-10} timeout -k 10 300 python scripts/ab_options.py codec "enc_segs=0" "enc_segs=1" "enc_segs=0" "enc_segs=1" 2>&1 | tail -4
CKPT=1024 ROUNDS=${ROUNDS:-10} timeout -k 10 300 python scripts/ab_options.py codec "enc_segs=0" "enc_segs=1" 2>&1 | tail -2
