#!/bin/bash
# tools/r05_traffic.sh <tag> name...: kernel time and memory-side traffic (FETCH_SIZE x 2 + WRITE_SIZE, two pmc passes) of configs[2]'s scene at 1024 spp per variant
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out/r06
{
for v in "$@"; do
  export KYHIP_LIB=$PWD/build_variants/$v.so
  python3 bench.py --workload veach --spp 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$v  %.1f Msamples/s  kernel %.2f ms' % (j['value'], j['roofline']['kernel_ms']))"
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C -d gpurun_out/tr_${v}_$C -o c -- python3 bench.py --workload veach --spp 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline > gpurun_out/tr_${v}_$C.log 2>&1
    python3 tools/rocprof_summary.py gpurun_out/tr_${v}_$C/c_results.db --pmc | grep -E "render_kernel.*$C" | awk -v v=$v -v c=$C '{printf "%s  %s  %.2f GB per launch (KiB counter x 1024%s)\n", v, c, $NF * 1024 * (c == "FETCH_SIZE" ? 2 : 1) / 1e9, (c == "FETCH_SIZE" ? " x 2" : "")}'
    rm -rf gpurun_out/tr_${v}_$C gpurun_out/tr_${v}_$C.log
  done
done
} > gpurun_out/r06/$TAG.txt 2>&1
cat gpurun_out/r06/$TAG.txt
