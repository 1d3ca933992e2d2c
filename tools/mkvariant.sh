#!/bin/bash
# tools/mkvariant.sh <name> [extra hipcc flags...]  -> build_variants/<name>.so + build_variants/<name>.txt (resource usage of the render kernels)
# The flags go to the translation units with kernels in them (ky_launch.hip, ky_kat.hip); the host-only objects come from build/obj (`make` first).
# -DKY_FEW_VARIANTS keeps the two headline kernels and one catch-all: 15 s instead of 90.
NAME=$1; shift
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -no-hip-rt -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt"
mkdir -p /tmp/bv build_variants
make -s build/obj/ky_pack.o build/obj/ky_jit.o build/obj/ky_seam.o || exit 1
hipcc $BASE "$@" -Rpass-analysis=kernel-resource-usage -c -o /tmp/bv/$NAME.launch.o ky_amd/csrc/ky_launch.hip 2> /tmp/bv/$NAME.log &
hipcc $BASE "$@" -c -o /tmp/bv/$NAME.kat.o ky_amd/csrc/ky_kat.hip 2> /tmp/bv/$NAME.kat.log &
wait
python3 tools/resources.py /tmp/bv/$NAME.log > build_variants/$NAME.txt
grep -E "error" /tmp/bv/$NAME.log /tmp/bv/$NAME.kat.log | head -5
if grep -q "error:" /tmp/bv/$NAME.log /tmp/bv/$NAME.kat.log; then echo "BUILD FAILED: $NAME"; exit 1; fi
HOSTOBJ="build/obj/ky_pack.o build/obj/ky_jit.o build/obj/ky_seam.o"
if [ -n "$KY_HOSTFLAGS" ]; then   # flags that change what host and device share (ky_shard.hpp's chunk schedule): the host objects are rebuilt with them too
  HOSTOBJ=""
  for f in ky_pack ky_jit ky_seam; do hipcc $BASE "$@" -c -o /tmp/bv/$NAME.$f.o ky_amd/csrc/$f.cpp || exit 1; HOSTOBJ="$HOSTOBJ /tmp/bv/$NAME.$f.o"; done
fi
hipcc --offload-arch=gfx950 -fPIC -no-hip-rt -shared -o build_variants/$NAME.so /tmp/bv/$NAME.launch.o /tmp/bv/$NAME.kat.o $HOSTOBJ || exit 1
echo "== $NAME: $*"; cat build_variants/$NAME.txt
