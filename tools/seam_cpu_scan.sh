#!/bin/bash
# boundary.rates[1] (configs[1] at 64 spp through kyhip_render) under CPU grants of 2, 4 and all CPUs, for 1 / 2 / 4 adding threads and the library's own choice
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
{
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for cpus in 0,1 0-3 all; do
  for t in auto 1 2 4; do
    if [ "$t" = auto ]; then unset KYHIP_SEAM_THREADS; else export KYHIP_SEAM_THREADS=$t; fi
    if [ "$cpus" = all ]; then python3 tools/seam_rate.py 2>/dev/null | tail -1; else taskset -c $cpus python3 tools/seam_rate.py 2>/dev/null | tail -1; fi
  done
done
} > gpurun_out/r05/seam_cpu_scan.txt 2>&1
cat gpurun_out/r05/seam_cpu_scan.txt
