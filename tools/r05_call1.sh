#!/bin/bash
# round 5, GPU call 1: baseline + first experiments (cmpx hit tests, plain flush, one-hash sampler) and the phase clocks at configs[1]'s own 1024 spp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
{
echo "== sweep"; tools/sweep.sh base e1 e12 e125 base
echo "== phase clocks, cornell 1024 spp"; KYHIP_LIB=$PWD/build_variants/clk.so python3 tools/phase_clocks.py 1024
echo "== phase clocks, cornell 64 spp"; KYHIP_LIB=$PWD/build_variants/clk.so python3 tools/phase_clocks.py 64
echo "== phase clocks, veach 1024 spp"; KYHIP_LIB=$PWD/build_variants/clk.so python3 tools/phase_clocks.py 1024 veach
echo "== icount"; tools/icount.sh base e12
} > gpurun_out/r05/call1.txt 2>&1
cat gpurun_out/r05/call1.txt
