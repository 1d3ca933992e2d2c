cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python3 bench.py "$@" 2>gpurun_out/r06/bline.err | tail -1 > gpurun_out/r06/bline.json; echo "rc=${PIPESTATUS[0]}"
python3 - <<'P'
import json
d = json.load(open("gpurun_out/r06/bline.json"))
for r in d["boundary"]["rates"]:
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != "statistic"})
print(d["value"], d["parity_gate"]["failed"], d["parity_gate"]["review"])
P
