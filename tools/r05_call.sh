#!/bin/bash
# round 5 A/B: tools/r05_call.sh <tag> name1 name2 ...   (sweep of build_variants/<name>.so on configs[1] and configs[2] at 512 spp, first name repeated at the end)
cd $GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out/r05
{ tools/sweep.sh "$@" $1; } > gpurun_out/r05/$TAG.txt 2>&1
cat gpurun_out/r05/$TAG.txt
