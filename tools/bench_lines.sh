#!/bin/bash
# the four bench lines of a round: tools/bench_lines.sh <prefix>  (run on the GPU box after profiles/valu.json was regenerated)
P=$1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_cornell.json
python3 bench.py --workload veach 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_veach.json
python3 bench.py --workload batch 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_batch.json
python3 bench.py --workload stress --steps 1 --warmup 0 2>/dev/null | tail -1 > gpurun_out/final/${P}_bench_line_stress.json
for w in cornell veach batch stress; do python3 -c "
import json
j=json.loads(open('gpurun_out/final/${P}_bench_line_$w.json').read())
print('$w', round(j['value'],1), 'ms/step', round(j['ms_per_step'],2), 'cpu', round(j['cpu_baseline']['value'],2), j['cpu_baseline']['cores'], 'rmse %.2e' % j['rmse_gpu_vs_cpu'])
"; done
