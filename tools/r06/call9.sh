cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/c9_tests.log 2>&1; tail -4 gpurun_out/r06/c9_tests.log
bash tools/final_profiles.sh r06_a > gpurun_out/r06/c9_final.log 2>&1; tail -5 gpurun_out/r06/c9_final.log
