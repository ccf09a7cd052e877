for r in 1 2; do for v in 49 41 33 25; do
env FGMM_EF_MIN_ROWS=$v FGMM_BENCH_DETAIL=/tmp/ab_detail.json python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sublegs > /tmp/ab_line.json 2>/dev/null
python - "ef_min=$v" <<'PY'
import json, sys, statistics as st
d = json.load(open("/tmp/ab_detail.json")); sm = d["step_ms"]; ph = sm.get("phases_ms") or {}
print(f"{sys.argv[1]:12s} value {d['value']:7.1f}  median {sm['median']:6.3f} p90 {sm['p90']:6.3f}  cpu_ms {st.median(sm['cpu_ms']):6.1f}  B/latent {d['pcie']['decode_table_bytes_per_latent']}  busy {ph.get('call1_decode.worker_busy')} {ph.get('call2_decode.worker_busy')}  bus {ph.get('call1_decode.bus')} tails {ph.get('call1_decode.host_tail')} {ph.get('call2_decode.host_tail')}", flush=True)
PY
done; done
