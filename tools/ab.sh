#!/bin/bash
# A/B bench of libkyhip builds under build_variants/: tools/ab.sh "<bench args>" name1 name2 ...
ARGS="$1"; shift
for v in "$@"; do
  echo "== $v"
  KYHIP_LIB=$PWD/build_variants/$v.so python bench.py $ARGS --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('  %.1f Msamples/s  kernel %.2f ms  frac %.4f  film_mean %.6f' % (j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['film_mean']))
"
done
