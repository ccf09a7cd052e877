for r in 1 2; do for v in 49 33; do
env FGMM_EF_MIN_ROWS=$v FGMM_BENCH_DETAIL=/tmp/ab_detail.json python bench.py --workload elic4k --images 8 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /tmp/ab_line.json 2>/dev/null
python - "elic ef_min=$v" <<'PY'
import json, sys, statistics as st
d = json.load(open("/tmp/ab_detail.json")); sm = d["step_ms"]
print(f"{sys.argv[1]:16s} value {d['value']:7.1f}  median {sm['median']:7.2f}  cpu_ms {st.median(sm['cpu_ms']):7.1f}  B/latent {d['pcie']['decode_table_bytes_per_latent']}", flush=True)
PY
done; done
for v in 0 6 10 12; do
env FGMM_PIECES=$v FGMM_BENCH_DETAIL=/tmp/ab_detail.json python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sublegs > /tmp/ab_line.json 2>/dev/null
python - "pieces=$v" <<'PY'
import json, sys, statistics as st
d = json.load(open("/tmp/ab_detail.json")); sm = d["step_ms"]; ph = sm.get("phases_ms") or {}
print(f"{sys.argv[1]:16s} value {d['value']:7.1f}  median {sm['median']:6.3f} p90 {sm['p90']:6.3f}  cpu_ms {st.median(sm['cpu_ms']):6.1f}  tails {ph.get('call1_decode.host_tail')} {ph.get('call2_decode.host_tail')}", flush=True)
PY
done
