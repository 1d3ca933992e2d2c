cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
tools/ab6.sh "--workload single --spp 512" envA envD envG envH envI envJ envA
} > gpurun_out/r06/call3.txt 2>&1
cat gpurun_out/r06/call3.txt
