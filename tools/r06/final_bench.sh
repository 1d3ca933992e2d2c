cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python3 bench.py > gpurun_out/r06/final_bench2.json 2> gpurun_out/r06/final_bench2.err; echo rc=$?
python3 -c "
import json
j = json.load(open('gpurun_out/r06/final_bench2.json'))
print(j['value'], j['parity_gate'])
print('headline', j['rmse_full_spp'])
for k, v in j['extra_workloads'].items():
    if isinstance(v, dict) and 'rmse_full_spp' in v: print(k, round(v['value']), v['rmse_full_spp'])
"
