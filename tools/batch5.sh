#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b5
python3 -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/b5/pytest.txt
python3 bench.py > gpurun_out/b5/bench_cornell.json 2> gpurun_out/b5/bench_cornell.err
python3 bench.py --workload veach > gpurun_out/b5/bench_veach.json 2> gpurun_out/b5/bench_veach.err
python3 bench.py --workload batch > gpurun_out/b5/bench_batch.json 2> gpurun_out/b5/bench_batch.err
python3 bench.py --workload stress --steps 1 --warmup 0 > gpurun_out/b5/bench_stress.json 2> gpurun_out/b5/bench_stress.err
KY_BENCH_ONE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --workload batch --spp 64 > gpurun_out/b5/bench_n2.json 2> gpurun_out/b5/bench_n2.err
tail -n 3 gpurun_out/b5/*.err; cat gpurun_out/b5/pytest.txt; for f in gpurun_out/b5/*.json; do echo $f; python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); print(' value %.1f ms/step %.2f frac %.3f kernel_ms %s' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_ms_per_frame'])); print(' cpu', j.get('cpu_baseline'), j.get('rmse_gpu_vs_cpu'))
"; done
