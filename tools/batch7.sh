#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b7
for q in 0 1; do for wl in cornell veach; do
  SPP=256; [ $wl = veach ] && SPP=128
  export KYHIP_SHADOW_QUEUE=$q
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d gpurun_out/b7/${wl}_q$q -o r -- python3 bench.py --workload $wl --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b7/${wl}_q$q.log 2>&1
  echo "== $wl queue=$q"; python3 tools/rocprof_summary.py gpurun_out/b7/${wl}_q$q/r_results.db --pmc | grep -E "render_kernel" | awk '{print $(NF-3), $(NF-2), $NF}' | tr '\n' ' '; echo
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVES -d gpurun_out/b7/${wl}_m$q -o r -- python3 bench.py --workload $wl --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b7/${wl}_m$q.log 2>&1
  python3 tools/rocprof_summary.py gpurun_out/b7/${wl}_m$q/r_results.db --pmc | grep -E "render_kernel" | awk '{print $(NF-3), $(NF-2), $NF}' | tr '\n' ' '; echo
done; done
