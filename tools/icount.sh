#!/bin/bash
# tools/icount.sh name1 name2 ...  : per-sample instruction counters of each build_variants/<name>.so on Cornell (C2) and Veach (C3 at 512 spp)
# (one rocprofv3 --pmc pass per variant and workload; prints VALU / SALU wave-instructions per camera sample, lane occupancy, wait fraction)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CNT="SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_LDS"
for v in "$@"; do
  for wl in "cornell" "veach --spp 512"; do
    T=${v}_$(echo $wl | cut -d' ' -f1)
    KYHIP_LIB=$PWD/build_variants/$v.so rocprofv3 --kernel-trace --pmc $CNT -d gpurun_out/ic_$T -o c -- python3 bench.py --no-cpu-baseline --no-extra --no-pipeline --steps 2 --warmup 1 --workload $wl > gpurun_out/ic_$T.log 2>&1
    python3 - "$T" gpurun_out/ic_$T/c_results.db "$wl" <<'PY'
import sqlite3, sys
tag, path, wl = sys.argv[1:4]
samples = 1024 * 768 * 1024 if wl.startswith("cornell") else 1280 * 720 * 512
db = sqlite3.connect(path)
c = {}
for name, n, tot in db.execute("select counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events where name like '%render_kernel%' group by counter_name"):
    c[name] = tot / max(n, 1)
g = lambda k: c.get(k, float('nan'))
valu = g("SQ_INSTS_VALU")
print("%-24s VALU/sample %7.2f  SALU/sample %6.2f  lanes %5.3f  wait %5.3f  vmem_wr/sample %6.3f  lds/sample %6.2f" % (
    tag, valu / samples, g("SQ_INSTS_SALU") / samples, g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")) if g("SQ_ACTIVE_INST_VALU") else float('nan'),
    g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_INSTS_VMEM_WR") / samples, g("SQ_INSTS_LDS") / samples))
PY
    rm -rf gpurun_out/ic_$T
  done
done
