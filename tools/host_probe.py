#!/usr/bin/env python3
"""What the host of the GPU box offers the CPU baseline: CPUs, affinity, cgroup quota, memory, NUMA nodes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(bench.host_cpus())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/memory.max", "/sys/devices/system/node/online",
          "/sys/kernel/mm/transparent_hugepage/enabled", "/proc/loadavg"):
    try:
        print(p, open(p).read().strip()[:200])
    except Exception as e:
        print(p, "-", e)
os.system("nproc; lscpu | head -25; free -g | head -3")
