cd "$(dirname "$0")/.."
ROUNDS=8 THREADS="16 0" bash scripts/host_threads_ab.sh
for rep in 1 2 3 4; do for t in 16 0; do echo -n "elic4k host threads $t: "; python bench.py --workload elic4k --steps 3 --warmup 1 --no-cpu-baseline --no-extras --host-threads $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], 'Mpix/s  ms/step', d['ms_per_step'], 'threads', d['config']['host_threads_per_gpu'], d['step_ms']['all'])"; done; done
