#!/bin/bash
# `make sanitize`: the CPU test suite under the address + undefined-behaviour builds of the host-only code (ky_pack.cpp, ky_jit.cpp, the host mirror,
# the oracle), then the two stress binaries (address / thread).  Writes build/san/*.log and a summary to profiles/<round>_sanitize_summary.txt.
# Never on the GPU box: the pool has no GPU sanitizers, and nothing here touches a device.
ROUND=${KY_ROUND:-r05}
cd "$(dirname "$0")/../.." || exit 1
ASAN=$(g++ -print-file-name=libasan.so); UBSAN=$(g++ -print-file-name=libubsan.so)
mkdir -p build/san profiles
export KYHIP_CACHE_DIR=$PWD/build/san/cache; rm -rf "$KYHIP_CACHE_DIR" build/san/cache_a build/san/cache_t
echo "== pytest -m 'not gpu' under LD_PRELOAD=libasan + libubsan (KY_SANITIZE=asan)"
# deselected: test_abi (the host-only library exports no GPU entry points by design) and the gloo multi-process tests (they import torch in child
# processes; torch under a preloaded ASan runtime takes minutes to import and its own allocator trips the leak checker)
KY_SANITIZE=asan LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
  timeout 3000 python3 -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_abi.py --ignore tests/test_dist_cpu.py > build/san/pytest_asan.log 2>&1
RC_PY=$?
tail -3 build/san/pytest_asan.log
echo "== stress, address + undefined"
KYHIP_CACHE_DIR=$PWD/build/san/cache_a KYHIP_HIPCC=$PWD/tools/sanitize/fake_hipcc.sh ASAN_OPTIONS=detect_leaks=1 build/san/stress_asan 3000 > build/san/stress_asan.log 2>&1; RC_A=$?
tail -5 build/san/stress_asan.log
echo "== stress, thread"
KYHIP_CACHE_DIR=$PWD/build/san/cache_t KYHIP_HIPCC=$PWD/tools/sanitize/fake_hipcc.sh TSAN_OPTIONS="halt_on_error=1 die_after_fork=0" build/san/stress_tsan 3000 > build/san/stress_tsan.log 2>&1; RC_T=$?
tail -5 build/san/stress_tsan.log
{
  echo "# make sanitize ($ROUND): g++ $(g++ -dumpversion), -fsanitize=address,undefined and -fsanitize=thread on the host-only code"
  echo "# sources: ky_amd/csrc/ky_pack.cpp ky_jit.cpp ky_hostcheck.cpp, ky_amd/host/ky_capi.cpp (ky.hpp), oracle/*.cpp; driver tools/sanitize/stress.cpp"
  echo "pytest -m 'not gpu' under ASan + UBSan: exit $RC_PY: $(tail -1 build/san/pytest_asan.log)"
  echo "  AddressSanitizer reports: $(grep -c 'ERROR: AddressSanitizer' build/san/pytest_asan.log)   UBSan runtime errors: $(grep -c 'runtime error:' build/san/pytest_asan.log)"
  echo "stress_asan 3000 (seam lock order x 2 threads x 2 devices with a fork in between, HostPool, banded add, chunk schedules, code cache x 6 threads): exit $RC_A"
  sed 's/^/  /' build/san/stress_asan.log | tail -6
  echo "  AddressSanitizer reports: $(grep -c 'ERROR: AddressSanitizer' build/san/stress_asan.log)   LeakSanitizer: $(grep -c 'ERROR: LeakSanitizer' build/san/stress_asan.log)   UBSan: $(grep -c 'runtime error:' build/san/stress_asan.log)"
  echo "stress_tsan 3000: exit $RC_T"
  sed 's/^/  /' build/san/stress_tsan.log | tail -6
  echo "  ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' build/san/stress_tsan.log)"
} > profiles/${ROUND}_sanitize_summary.txt
cat profiles/${ROUND}_sanitize_summary.txt
[ $RC_PY -eq 0 ] && [ $RC_A -eq 0 ] && [ $RC_T -eq 0 ]
