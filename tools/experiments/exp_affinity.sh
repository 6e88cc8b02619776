#!/bin/bash
# Does thread placement matter for the host-bound rate?  (the cgroup quota is CPU time, not a cpuset:
# 48 lane threads float over 256 hardware threads of two sockets)
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA" 
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) local_cpulist=$(cat $d/local_cpulist 2>/dev/null)"; done 2>/dev/null | head -8
cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list /sys/devices/system/cpu/cpu1/topology/thread_siblings_list
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-output"
q() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2', d['value'], 'fps  cpu ms/pic', d['host_cpu_ms_per_picture'], 'busy', d['host_cpus_busy_rank0'])"; }
$B > gpurun_out/aff_none.json 2>/dev/null; q gpurun_out/aff_none.json "no affinity      "
for spec in "0-63" "64-127" "0-31" "0-15" "0-63,128-191" "0-127"; do
  taskset -c $spec $B > gpurun_out/aff.json 2>/dev/null; q gpurun_out/aff.json "taskset $spec"
done
$B > gpurun_out/aff_none2.json 2>/dev/null; q gpurun_out/aff_none2.json "no affinity again "
