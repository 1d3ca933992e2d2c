cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06; python3 bench.py --no-extra > gpurun_out/r06/hp.json 2> gpurun_out/r06/hp.err; echo rc=$?
python3 -c "
import json
j = json.load(open('gpurun_out/r06/hp.json'))
print(j['rmse_full_spp'])
"
