#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b8
export KYHIP_SHADOW_QUEUE=1
tools/sweep.sh sqa0 sqa1 sqa2 sqa3 sqa4 > gpurun_out/b8/sweep.txt 2>&1
KYHIP_SHADOW_QUEUE=0 tools/sweep.sh sqa0 >> gpurun_out/b8/sweep.txt 2>&1
cat gpurun_out/b8/sweep.txt
