#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b2
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/b2/pytest.txt
./build_variants/valu_peak > gpurun_out/b2/valu_peak.txt 2>&1
tools/sweep.sh base defer dit dit_r2 dit_r3 dit_t24 dit_r2t56 dit_c64 dit_c128 base > gpurun_out/b2/sweep.txt 2>&1
for s in cornell veach; do KYHIP_LIB=$PWD/build_variants/lanes2.so python3 tools/lane_probe.py $s > gpurun_out/b2/lanes_$s.txt 2>&1; done
cat gpurun_out/b2/*.txt
