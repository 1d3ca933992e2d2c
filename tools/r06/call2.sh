cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
tools/ab6.sh "--workload single --spp 512" envA envB envC envD envE envF envA
tools/ab6.sh "--workload cornell" base trX trY base
} > gpurun_out/r06/call2.txt 2>&1
cat gpurun_out/r06/call2.txt
