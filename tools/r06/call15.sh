cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
bash tools/final_profiles.sh r06_e > gpurun_out/r06/c15_final.log 2>&1; tail -3 gpurun_out/r06/c15_final.log
