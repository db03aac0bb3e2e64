#!/bin/bash
# Sanitizer passes over the HOST half of the library (untrusted .pfv bytes parsed on worker threads; C++ where the reference has safe Rust,
# SURVEY section 5): the CPU-emulator build of the unmodified csrc/ (tests/hipemu) under AddressSanitizer + UndefinedBehaviorSanitizer and
# under ThreadSanitizer, running tools/sanitize_run.py (the damaged-stream, GOP-object, device-entropy and batch-object checks of the CPU
# suite, without pytest: its process handling hangs under ThreadSanitizer's runtime).  The emulator's fibers are announced to TSan
# (tests/hipemu/hipemu.cpp: __tsan_switch_to_fiber).  TSan runs one process per step: a process that has created and joined several hundred
# pool threads AND switched fibers a few million times dies inside the TSan runtime (SEGV in its own shadow, no report) -- every step alone passes.
# Logs: profiles/r05_sanitize_{asan_ubsan,tsan}.log       usage: bash tools/sanitize.sh [asan|tsan|all]   (no GPU needed; `make sanitize`)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
WHAT=${1:-all}
GCC_LIBDIR=$(dirname "$(gcc -print-file-name=libasan.so)")
STEPS=("corrupted streams" "look-ahead reset" "stream round trip" "GOP objects" "damaged streams" "device entropy" "batch encoder")
count() { grep -c -E 'ERROR: (Address|Thread|Leak)Sanitizer|WARNING: ThreadSanitizer|runtime error:' "$1"; }
rc=0
if [ "$WHAT" = asan ] || [ "$WHAT" = all ]; then
  log=profiles/r05_sanitize_asan_ubsan.log
  flags="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
  echo "== asan_ubsan: g++ $flags (tests/conftest.py build_emulator, PFV_EMU_DEFS); tools/sanitize_run.py -- $(date -u +%FT%TZ), $(gcc --version | head -1)" > $log
  PFV_EMU_DEFS="$flags" LD_PRELOAD="$GCC_LIBDIR/libasan.so" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 timeout 3000 python tools/sanitize_run.py >> $log 2>&1 || rc=1
  echo "== exit code $rc; sanitizer reports in this log: $(count $log)" >> $log; tail -2 $log
fi
if [ "$WHAT" = tsan ] || [ "$WHAT" = all ]; then
  log=profiles/r05_sanitize_tsan.log
  flags="-fsanitize=thread -fno-omit-frame-pointer"
  echo "== tsan: g++ $flags (tests/conftest.py build_emulator, PFV_EMU_DEFS); tools/sanitize_run.py, one process per step -- $(date -u +%FT%TZ), $(gcc --version | head -1)" > $log
  for s in "${STEPS[@]}"; do
    # gcc 11's libtsan preloaded into python hangs at start-up every other process here (no CPU, no output): a step that has printed nothing
    # after 150 s is stopped (its own PID) and started again, up to 6 times; one that is running gets 900 s
    done_step=0
    for attempt in 1 2 3 4 5 6; do
      tmp=$(mktemp)
      PFV_SAN_ONLY="$s" PFV_EMU_DEFS="$flags" LD_PRELOAD="$GCC_LIBDIR/libtsan.so" TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 report_signal_unsafe=0" timeout 900 python tools/sanitize_run.py > $tmp 2>&1 &
      pid=$!
      for i in $(seq 1 30); do sleep 5; [ -s $tmp ] && break; kill -0 $pid 2>/dev/null || break; done
      if [ ! -s $tmp ] && kill -0 $pid 2>/dev/null; then kill $pid; wait $pid 2>/dev/null; echo "== step '$s': attempt $attempt hung at start-up (no output after 150 s), stopped" >> $log; rm -f $tmp; continue; fi
      if wait $pid; then done_step=1; fi
      cat $tmp >> $log; rm -f $tmp
      break
    done
    [ $done_step = 1 ] || { rc=1; echo "== step '$s' did not finish" >> $log; }
  done
  echo "== exit code $rc; sanitizer reports in this log: $(count $log)" >> $log; tail -2 $log
fi
exit $rc
