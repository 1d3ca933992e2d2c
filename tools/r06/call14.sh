cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06 gpurun_out/final
P=r06_d
python3 tools/grid_rates.py 256 2>/dev/null > gpurun_out/final/${P}_grid_rates.txt
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_cornell.json
python3 bench.py --workload veach --no-extra 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_veach.json
python3 bench.py --workload batch --no-extra 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_batch.json
python3 bench.py --workload single --no-extra 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_single.json
ls -la gpurun_out/final | grep r06_d | tail -6
bash tools/r06/soak.sh
