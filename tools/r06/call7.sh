cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
tools/ab6.sh "--workload veach --spp 512" vchA vchB vchA
tools/ab6.sh "--workload cornell" vchA vchB
} > gpurun_out/r06/call7.txt 2>&1
cat gpurun_out/r06/call7.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r06/c7_tests.log 2>&1; tail -15 gpurun_out/r06/c7_tests.log
