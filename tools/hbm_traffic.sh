#!/bin/bash
# HBM traffic of render_kernel from the TCC counters, two separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tools/hbm_traffic.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/hbm_${TAG}_$C -o $C -- python3 bench.py --no-cpu-baseline --no-extra --no-pipeline "$@" > gpurun_out/hbm_${TAG}_$C.log 2>&1
  python3 tools/rocprof_summary.py gpurun_out/hbm_${TAG}_$C/${C}_results.db --pmc | grep -E "render_kernel.*$C"
done
