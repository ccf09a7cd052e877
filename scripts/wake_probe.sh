#!/usr/bin/env bash
# default slice against a 0.1 ms slice, interleaved, unbound and on the second NUMA node   -> gpurun_out/wake_probe.txt
mkdir -p gpurun_out
{
  uname -r; cat /proc/loadavg; grep -E "^some" /proc/pressure/cpu
  node1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
  for r in 1 2 3 4; do for s in 0 100 1000; do
    ./scripts/bin/wake_probe 16 2 $s
    [ -n "$node1" ] && { echo -n "node1: "; taskset -c $node1 ./scripts/bin/wake_probe 16 2 $s; }
  done; done
  cat /proc/loadavg
} > gpurun_out/wake_probe.txt 2>&1
echo done
