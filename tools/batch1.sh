#!/bin/bash
# round-2 first GPU batch: sanity + probes + compiler flag sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b1
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/b1/pytest.txt
./build_variants/valu_rate > gpurun_out/b1/valu_rate.txt 2>&1
for s in cornell veach; do KYHIP_LIB=$PWD/build_variants/lanes.so python3 tools/lane_probe.py $s > gpurun_out/b1/lanes_$s.txt 2>&1; done
KYHIP_LIB=$PWD/build_variants/clocks.so python3 tools/phase_clocks.py 64 > gpurun_out/b1/clocks_cornell.txt 2>&1
KYHIP_LIB=$PWD/build_variants/clocks.so python3 tools/phase_clocks.py 64 veach > gpurun_out/b1/clocks_veach.txt 2>&1
tools/sweep.sh base ilp memcl trackers bias100 wprio ifcvt base > gpurun_out/b1/sweep.txt 2>&1
cat gpurun_out/b1/*.txt
