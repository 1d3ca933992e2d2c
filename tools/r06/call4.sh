cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
tools/ab6.sh "--workload single --spp 512" envJ envK envL envM envN envO envJ
} > gpurun_out/r06/call4.txt 2>&1
cat gpurun_out/r06/call4.txt
