cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/c11_tests.log 2>&1; tail -6 gpurun_out/r06/c11_tests.log
