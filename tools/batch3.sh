#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b3
tools/sweep.sh d0 bk2 bk4 bk8 bk4l4 bk4c64 bk16l16 d0 > gpurun_out/b3/sweep.txt 2>&1
cat gpurun_out/b3/*.txt
