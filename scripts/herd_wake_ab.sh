#!/usr/bin/env bash
# wake_probe (16 sleep/wake threads) beside a herd of 120 threads that all run 3 ms every 100 ms, on the second NUMA node's CPUs:
# default slice against short slices   -> gpurun_out/herd_wake_ab.txt
mkdir -p gpurun_out
node1=$(cat /sys/devices/system/node/node1/cpulist)
{
  cat /proc/loadavg
  echo "--- no herd"; taskset -c $node1 ./scripts/bin/wake_probe 16 2 0
  taskset -c $node1 ./scripts/bin/herd 120 3000 40 &
  sleep 0.3
  for r in 1 2 3; do for s in 0 100 500; do echo -n "herd: "; taskset -c $node1 ./scripts/bin/wake_probe 16 2 $s; done; done
  kill %1; wait
  cat /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
} > gpurun_out/herd_wake_ab.txt 2>&1
echo done
