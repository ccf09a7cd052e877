#!/usr/bin/env bash
# Is the 100-ms-cadence stall the kernel's automatic NUMA balancing scanning the process?  The same diagnostic leg without / with an
# explicit memory policy (FGMM_BENCH_MEMPOLICY=1: set_mempolicy(MPOL_PREFERRED, the GPU's node) before any thread is created).
out=gpurun_out/numa_ab; mkdir -p $out
{ echo "numa_balancing = $(cat /proc/sys/kernel/numa_balancing 2>&1)"; grep -E "^numa_(pte_updates|hint_faults|pages_migrated)" /proc/vmstat; } > $out/sys.txt
for rep in 1 2; do for mp in 0 1; do
  FGMM_BENCH_MEMPOLICY=$mp python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sublegs --diag-steps 80 --diag-configs ${CFGS:-48:0,48:100,16:0,16:100} > $out/bench_mp${mp}_$rep.json 2> $out/bench_mp${mp}_$rep.err || { tail -5 $out/bench_mp${mp}_$rep.err; exit 1; }
  { echo "after mp=$mp rep=$rep"; grep -E "^numa_(pte_updates|hint_faults|pages_migrated)" /proc/vmstat; } >> $out/sys.txt
done; done
cat $out/sys.txt
