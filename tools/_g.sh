KYHIP_LIB=$PWD/build_variants/clk.so python tools/phase_clocks.py 256 2>&1 | grep -v amdgpu
KYHIP_LIB=$PWD/build_variants/lanes.so python tools/lane_probe.py cornell 2>&1 | grep -v amdgpu
