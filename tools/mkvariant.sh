#!/bin/bash
# tools/mkvariant.sh <name> [extra hipcc flags...]  -> build_variants/<name>.so + build_variants/<name>.txt (resource usage of the render kernels)
NAME=$1; shift
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -no-hip-rt -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt"
SRC=${KY_SRC:-ky_amd/csrc/kyhip.hip}
mkdir -p /tmp/bv build_variants
make -s ky_amd/csrc/ky_rtc_sources.inc; hipcc $BASE "$@" -Rpass-analysis=kernel-resource-usage -shared -o build_variants/$NAME.so $SRC 2> /tmp/bv/$NAME.log
python3 tools/resources.py /tmp/bv/$NAME.log > build_variants/$NAME.txt
grep -E "error" /tmp/bv/$NAME.log | head -5
if grep -q "error:" /tmp/bv/$NAME.log; then echo "BUILD FAILED: $NAME"; exit 1; fi
echo "== $NAME: $*"; cat build_variants/$NAME.txt
