#!/bin/bash
# Regenerates the judged profile set of a round on the GPU box: tools/final_profiles.sh <prefix, e.g. r02_c>   (after `make all ubench` here)
# For each of five kernels -- the both_mis instantiation for one rectangle light (Cornell, configs[1]), the one with deferred shadow
# rays (Veach, configs[2]), the light_mis instantiation, the run-time-dispatched render_kernel<false,-1> (both Cornell, light_mis) and
# path_tracing_recursion_t's instantiation (Cornell) -- five summaries: kernel trace + stats, SQ issue counters, SQ instruction mix,
# FETCH_SIZE and WRITE_SIZE in separate pmc passes.  Then the per-cell rates of ky's experiment grids, the shard scan and the bench lines.
P=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python3 -c "from ky_amd import _abi as A; print('%016x' % A.load_kyhip().kyhip_kernel_source_hash())" > gpurun_out/final/${P}_kernel_source_hash.txt
run() {  # tag, bench args...
  T=$1; shift
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${P}_$T -o s -- python3 bench.py --no-cpu-baseline --no-extra --no-pipeline "$@" > gpurun_out/final/${P}_${T}_stats.log 2>&1
  python3 tools/rocprof_summary.py gpurun_out/prof_${P}_$T/s_results.db > gpurun_out/final/${P}_${T}_kernel_stats.txt
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C -d gpurun_out/prof_${P}_${T}_$C -o c -- python3 bench.py --no-cpu-baseline --no-extra --no-pipeline "$@" > gpurun_out/final/${P}_${T}_$C.log 2>&1
    python3 tools/rocprof_summary.py gpurun_out/prof_${P}_${T}_$C/c_results.db --pmc > gpurun_out/final/${P}_${T}_hbm_$(echo $C | tr A-Z a-z).txt
  done
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d gpurun_out/prof_${P}_${T}_issue -o i -- python3 bench.py --no-cpu-baseline --no-extra --no-pipeline --steps 2 "$@" > gpurun_out/final/${P}_${T}_issue.log 2>&1
  python3 tools/rocprof_summary.py gpurun_out/prof_${P}_${T}_issue/i_results.db --pmc > gpurun_out/final/${P}_${T}_pmc_sq_issue.txt
  rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d gpurun_out/prof_${P}_${T}_mix -o m -- python3 bench.py --no-cpu-baseline --no-extra --no-pipeline --steps 2 "$@" > gpurun_out/final/${P}_${T}_mix.log 2>&1
  python3 tools/rocprof_summary.py gpurun_out/prof_${P}_${T}_mix/m_results.db --pmc > gpurun_out/final/${P}_${T}_pmc_sq_mix.txt
}
run cornell --workload cornell
run veach --workload veach --spp 1024
run light_mis --workload cornell --direct-sample 32
export KYHIP_SPECIALISE=0   # the run-time-dispatched kernel (what the recursive integrators and general scenes run on), on the same workload
run generic --workload cornell --direct-sample 32
unset KYHIP_SPECIALISE
run recursion --workload cornell --integrator 9   # one of the recursive integrators on its own instantiation (render_multiple_integrator's cells)
run stress --workload stress --spp 256 --steps 2   # configs[4]'s geometry at 1/64 of its spp (round 5: its own counter set)
run batch --workload batch --spp 256 --steps 2     # configs[3]'s six frames at 1/8 of their spp (round 5: its own counter set, summed over the step's kernels)
run single --workload single --spp 512 --steps 2   # ky's own default driver (render_single_scene: the Cornell box under the environment light) at 1/4 of the bench's spp (round 6)
./build_variants/valu_peak > gpurun_out/final/${P}_valu_peak_ubench.txt 2>&1
./build_variants/valu_pk > gpurun_out/final/${P}_valu_pk_ubench.txt 2>&1
./build_variants/salu_mix > gpurun_out/final/${P}_salu_mix_ubench.txt 2>&1
python3 tools/grid_rates.py 256 2>/dev/null > gpurun_out/final/${P}_grid_rates.txt
python3 tools/shard_scan.py 2>/dev/null > gpurun_out/final/${P}_shard_scan.txt
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_cornell.json
python3 bench.py --workload veach --no-extra 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_veach.json
python3 bench.py --workload batch --no-extra 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_batch.json
python3 bench.py --workload stress --steps 1 --warmup 0 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_stress.json
python3 bench.py --workload single --no-extra 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_single.json
python3 bench.py --workload cornell --direct-sample 32 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_cornell_light_mis.json
rm -rf gpurun_out/prof_${P}_*   # the raw databases: 3 MB per pass, and gpurun brings home at most 64 MiB
ls -la gpurun_out/final | tail -40
