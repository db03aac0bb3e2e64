#!/bin/bash
# Sanitizer passes over the HOST half of the library (untrusted .pfv bytes parsed on worker threads; C++ where the reference has safe Rust,
# SURVEY section 5): the CPU-emulator build of the unmodified csrc/ (tests/hipemu) under AddressSanitizer + UndefinedBehaviorSanitizer and
# under ThreadSanitizer, running the damaged-stream, GOP-object and batch-decoder tests.  Logs: profiles/r05_sanitize_{asan_ubsan,tsan}.log
#   usage: bash tools/sanitize.sh [asan|tsan|all]        (no GPU needed; `make sanitize`)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
WHAT=${1:-all}
TESTS='corrupted_streams or gop_ or batch_decoder or stream_roundtrip or lists_decode'
GCC_LIBDIR=$(dirname "$(gcc -print-file-name=libasan.so)")
run() {  # name, compile flags, preload library, environment
  name=$1; flags=$2; lib=$3; shift 3
  log=profiles/r05_sanitize_$name.log
  { echo "== $name: g++ $flags (tests/conftest.py build_emulator, PFV_EMU_DEFS) -- $(date -u +%FT%TZ), $(gcc --version | head -1)"
    echo "== tests: -k \"$TESTS\" of tests/test_emulated_kernels.py"; } > $log
  env "$@" PFV_EMU_DEFS="$flags" LD_PRELOAD="$GCC_LIBDIR/$lib" timeout 3000 python -m pytest tests/test_emulated_kernels.py -x -q -s -p no:cacheprovider -k "$TESTS" >> $log 2>&1
  rc=$?
  echo "== exit code $rc; sanitizer reports in this log: $(grep -c -E 'ERROR: (Address|Thread|Leak)Sanitizer|WARNING: ThreadSanitizer|runtime error:' $log)" >> $log
  tail -3 $log
  return $rc
}
rc=0
if [ "$WHAT" = asan ] || [ "$WHAT" = all ]; then
  run asan_ubsan "-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined" libasan.so ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 || rc=1
fi
if [ "$WHAT" = tsan ] || [ "$WHAT" = all ]; then
  run tsan "-fsanitize=thread -fno-omit-frame-pointer" libtsan.so TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 report_signal_unsafe=0" || rc=1
fi
exit $rc
