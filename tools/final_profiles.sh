#!/bin/bash
# Regenerates the judged profile set of a round on the GPU box: tools/final_profiles.sh <prefix, e.g. r01_f>
# (kernel trace + stats, SQ issue / mix counters, HBM traffic in two separate pmc passes, the bench line itself).
P=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$P -o $P -- python3 bench.py --no-cpu-baseline > gpurun_out/final/${P}_bench_under_rocprof.log 2>&1
python3 tools/rocprof_summary.py gpurun_out/prof_$P/${P}_results.db > gpurun_out/final/${P}_final_kernel_stats.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/hbm_${P}_$C -o $C -- python3 bench.py --no-cpu-baseline > gpurun_out/final/${P}_hbm_$C.log 2>&1
  python3 tools/rocprof_summary.py gpurun_out/hbm_${P}_$C/${C}_results.db --pmc > gpurun_out/final/${P}_hbm_$(echo $C | tr A-Z a-z).txt
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d gpurun_out/pmc_${P}_issue -o issue -- python3 bench.py --no-cpu-baseline --steps 2 > gpurun_out/final/${P}_pmc_issue.log 2>&1
python3 tools/rocprof_summary.py gpurun_out/pmc_${P}_issue/issue_results.db --pmc > gpurun_out/final/${P}_pmc_sq_issue.txt
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d gpurun_out/pmc_${P}_mix -o mix -- python3 bench.py --no-cpu-baseline --steps 2 > gpurun_out/final/${P}_pmc_mix.log 2>&1
python3 tools/rocprof_summary.py gpurun_out/pmc_${P}_mix/mix_results.db --pmc > gpurun_out/final/${P}_pmc_sq_mix.txt
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line.json
python3 bench.py --workload veach --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_veach.json
ls -la gpurun_out/final
