cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
tools/ab6.sh "--workload veach --spp 512" vchB vchC vchD vchE vchB
} > gpurun_out/r06/call8.txt 2>&1
cat gpurun_out/r06/call8.txt
