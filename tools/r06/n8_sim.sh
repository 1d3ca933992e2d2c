# eight ranks of bench.py on ONE GPU over gloo (KY_BENCH_ONE_GPU=1): the N = 8 code path end to end; the numbers mean nothing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
KY_BENCH_ONE_GPU=1 timeout 1300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r06/n8.out 2> gpurun_out/r06/n8.err
echo rc=$?
tail -1 gpurun_out/r06/n8.out | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['value'], j['n_gpus'], j['ms_per_step'], j['communicator'], j['film_mean'], j['launch_mode'], len(j['ranks']))"
tail -3 gpurun_out/r06/n8.err
