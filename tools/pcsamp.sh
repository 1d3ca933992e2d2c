#!/bin/bash
# PC sampling of the render kernel (beta): tools/pcsamp.sh <tag> <unit> <method> <interval> [bench args]
TAG=$1; UNIT=$2; METHOD=$3; IVL=$4; shift 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $IVL --output-format csv -d gpurun_out/pcs_$TAG -o $TAG -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/pcs_$TAG.log 2>&1
ls -la gpurun_out/pcs_$TAG 2>/dev/null | head; grep -E "^E|error|not supported" gpurun_out/pcs_$TAG.log | head -3
