#!/bin/bash
# tools/asm.sh <name> [extra hipcc flags...] -> build_variants/<name>.s: device assembly of ky_launch.hip (KY_SRC overrides the source)
NAME=$1; shift
BASE="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-bitwise-instead-of-logical -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt"
SRC=${KY_SRC:-ky_amd/csrc/ky_launch.hip}
mkdir -p /tmp/kyasm
hipcc $BASE "$@" -S --cuda-device-only -gline-tables-only -o /tmp/kyasm/$NAME.s $SRC
