#!/bin/bash
# VALU instructions of each phase of the lane engine by ablation: builds with -DKY_ABL=n keep the paths and the random streams
# but drop one piece of the direct-lighting code; the drop in SQ_INSTS_VALU is that piece's dynamic cost.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/abl
for wl in cornell veach; do
  for v in abl0 abl1 abl2 abl3 abl4 abl7; do
    export KYHIP_LIB=$PWD/build_variants/$v.so
    SPP=256; [ $wl = veach ] && SPP=128
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d gpurun_out/abl/${wl}_$v -o r -- python3 bench.py --workload $wl --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/abl/${wl}_$v.log 2>&1
    echo "== $wl $v"; python3 tools/rocprof_summary.py gpurun_out/abl/${wl}_$v/r_results.db --pmc | grep -E "render_kernel" | awk '{print $(NF-3), $(NF-2), $NF}' | tr '\n' ' '; echo
  done
done
