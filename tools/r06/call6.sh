cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/c6_tests.log 2>&1; tail -15 gpurun_out/r06/c6_tests.log
python bench.py > gpurun_out/r06/c6_bench.json 2> gpurun_out/r06/c6_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r06/c6_bench.json; tail -5 gpurun_out/r06/c6_bench.err
