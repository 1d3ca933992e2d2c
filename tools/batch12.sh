#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b12
tools/sweep.sh q6 q5 q5r2 > gpurun_out/b12/sweep.txt 2>&1
for v in q6 g6; do KYHIP_LIB=$PWD/build_variants/$v.so python3 tools/generic_time.py >> gpurun_out/b12/sweep.txt 2>&1; done
cat gpurun_out/b12/sweep.txt
