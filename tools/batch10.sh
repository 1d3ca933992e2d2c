#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b10
export KYHIP_SHADOW_QUEUE=0
for wl in cornell veach; do
  SPP=256; [ $wl = veach ] && SPP=128
  rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d gpurun_out/b10/${wl}_a -o r -- python3 bench.py --workload $wl --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b10/${wl}_a.log 2>&1
  echo "== $wl"; python3 tools/rocprof_summary.py gpurun_out/b10/${wl}_a/r_results.db --pmc | grep -E "render_kernel" | awk '{print $(NF-3), $(NF-2), $NF}' | tr '\n' ' '; echo
  rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_CYCLES -d gpurun_out/b10/${wl}_b -o r -- python3 bench.py --workload $wl --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b10/${wl}_b.log 2>&1
  python3 tools/rocprof_summary.py gpurun_out/b10/${wl}_b/r_results.db --pmc | grep -E "render_kernel" | awk '{print $(NF-3), $(NF-2), $NF}' | tr '\n' ' '; echo
done
