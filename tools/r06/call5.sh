cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
tools/ab6.sh "--workload single --spp 512" envJ envP envQ envR envS envT envJ
} > gpurun_out/r06/call5.txt 2>&1
cat gpurun_out/r06/call5.txt
