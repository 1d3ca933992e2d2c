cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/c16_tests.log 2>&1; tail -3 gpurun_out/r06/c16_tests.log
bash tools/final_profiles.sh r06_e > gpurun_out/r06/c16_final.log 2>&1; tail -3 gpurun_out/r06/c16_final.log
