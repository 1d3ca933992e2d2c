#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b9
KYHIP_SHADOW_QUEUE=1 tools/sweep.sh sqw5 sqw4 > gpurun_out/b9/sweep.txt 2>&1
KYHIP_SHADOW_QUEUE=0 tools/sweep.sh sqw5 sqw4 sqa0 >> gpurun_out/b9/sweep.txt 2>&1
cat gpurun_out/b9/sweep.txt
