#!/bin/bash
# tools/pmc.sh <tag> <lib variant or ""> "<counters>" [bench args]   (PMC pass: kernel-trace + pmc only)
TAG=$1; VAR=$2; CNT=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$VAR" ] && export KYHIP_LIB=$PWD/build_variants/$VAR.so
rocprofv3 --kernel-trace --pmc $CNT -d gpurun_out/pmc_$TAG -o $TAG -- python3 bench.py --no-cpu-baseline --no-extra --no-pipeline "$@" > gpurun_out/pmc_$TAG.log 2>&1
python3 tools/rocprof_summary.py gpurun_out/pmc_$TAG/${TAG}_results.db --pmc | grep -E "render_kernel|counter" 
