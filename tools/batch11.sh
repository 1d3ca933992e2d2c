#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b11
python3 -m pytest tests -m gpu -q -x 2>&1 | tail -3 > gpurun_out/b11/pytest.txt
KYHIP_SHADOW_QUEUE=0 tools/sweep.sh sqa0 ng > gpurun_out/b11/sweep.txt 2>&1
KYHIP_SHADOW_QUEUE=1 tools/sweep.sh ng >> gpurun_out/b11/sweep.txt 2>&1
cat gpurun_out/b11/*.txt
