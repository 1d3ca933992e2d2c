cd $GRAFT_REPO_ROOT
for B in 1 0; do
KYHIP_BOXES=$B python3 bench.py --workload batch --steps 1 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('batch boxes=$B %8.1f Msamples/s  frames ms' % j['value'], [round(x,1) for x in j['roofline']['kernel_ms_per_frame']])
"
done
