cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
tools/ab6.sh "--workload veach --spp 512" vchB vchC2 vchB
tools/ab6.sh "--workload cornell" full8 full7 full8
tools/ab6.sh "--workload batch --spp 256" full8 full7
} > gpurun_out/r06/call10.txt 2>&1
cat gpurun_out/r06/call10.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r06/c10_tests.log 2>&1; tail -4 gpurun_out/r06/c10_tests.log
