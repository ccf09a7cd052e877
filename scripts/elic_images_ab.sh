cd "$(dirname "$0")/.."
for imgs in 8 16; do for cap in 12288 8192; do
echo -n "elic4k images $imgs cap_e $cap: "
FGMM_TAB_CAP_E=$cap python bench.py --workload elic4k --images $imgs --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], 'Mpix/s  ms/step', d['ms_per_step'], ' upper', d['upper_bound']['value'], ' tab_ms', d['roofline_decode']['ms_per_step'], 'B/latent', d['pcie']['decode_table_bytes_per_latent'], ' ckpt', d['checkpointed']['value'], 'symtab frac', d['roofline']['frac'])"
done; done
