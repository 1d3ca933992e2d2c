cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/final_tests.log 2>&1; tail -4 gpurun_out/r06/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/final_bench.json 2> gpurun_out/r06/final_bench.err; echo "bench rc=$?"; tail -c 400 gpurun_out/r06/final_bench.json
