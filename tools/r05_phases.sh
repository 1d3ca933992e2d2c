#!/bin/bash
# round 5: phase clocks and lane probes of the final kernels (build_variants/clk.so, lanes.so: tools/mkvariant.sh clk -DKY_FEW_VARIANTS -DKY_PROFILE_CLOCKS, lanes ... -DKY_PROFILE_LANES)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
{
echo "== phase clocks cornell 1024"; KYHIP_LIB=$PWD/build_variants/clk.so python3 tools/phase_clocks.py 1024 2>/dev/null
echo "== phase clocks veach 1024"; KYHIP_LIB=$PWD/build_variants/clk.so python3 tools/phase_clocks.py 1024 veach 2>/dev/null
echo "== lanes cornell"; KYHIP_LIB=$PWD/build_variants/lanes.so python3 tools/lane_probe.py 2>/dev/null
echo "== lanes veach"; KYHIP_LIB=$PWD/build_variants/lanes.so python3 tools/lane_probe.py veach 2>/dev/null
} > gpurun_out/r05/phases_lanes_final.txt 2>&1
cat gpurun_out/r05/phases_lanes_final.txt
