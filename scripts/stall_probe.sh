#!/usr/bin/env bash
# who stalls the box's threads every ~100 ms?  (no GPU work)   -> gpurun_out/stall_probe.txt
out=gpurun_out/stall_probe.txt
mkdir -p gpurun_out
{
  echo "# $(date) $(uname -r) nproc $(nproc)"; cat /proc/self/cgroup
  for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/*/cpu.max /sys/fs/cgroup/*/*/cpu.max /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us /proc/sys/kernel/sched_cfs_bandwidth_slice_us; do [ -r $f ] && echo "$f: $(cat $f)"; done
  for n in /sys/devices/system/node/node*/cpulist; do echo "$n: $(cat $n)"; done
  grep -E "^(cpu |procs_running|procs_blocked)" /proc/stat
  cat /proc/loadavg
  for n in 1 12 24 48; do echo "=== unbound, $n spinners"; ./scripts/bin/stall_probe $n 1.2; done
  node1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
  if [ -n "$node1" ]; then for n in 1 12 48; do echo "=== taskset -c $node1, $n spinners"; taskset -c $node1 ./scripts/bin/stall_probe $n 1.2; done; fi
  cat /proc/loadavg
} > $out 2>&1
echo done
