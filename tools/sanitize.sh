#!/bin/bash
# Sanitizer passes over the HOST half of the library (untrusted .pfv bytes parsed on worker threads; C++ where the reference has safe Rust,
# SURVEY section 5): the CPU-emulator build of the unmodified csrc/ (tests/hipemu) under AddressSanitizer + UndefinedBehaviorSanitizer and
# under ThreadSanitizer.  ASan + UBSan run tools/sanitize_run.py (the damaged-stream, GOP-object, device-entropy and batch-object checks of the CPU
# suite, without pytest).  TSan runs the C++ mirror's round-trip program natively (tools/sanitize_native.py: python with a preloaded libtsan
# hangs at start-up here); the emulator's fibers are announced to TSan (tests/hipemu/hipemu.cpp: __tsan_switch_to_fiber).
# Logs: profiles/r06_sanitize_{asan_ubsan,tsan}.log       usage: bash tools/sanitize.sh [asan|tsan|all]   (no GPU needed; `make sanitize`)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
WHAT=${1:-all}
GCC_LIBDIR=$(dirname "$(gcc -print-file-name=libasan.so)")
STEPS=("corrupted streams" "look-ahead reset" "stream round trip" "GOP objects" "damaged streams" "device entropy" "batch encoder")
count() { grep -c -E 'ERROR: (Address|Thread|Leak)Sanitizer|WARNING: ThreadSanitizer|runtime error:' "$1"; }
rc=0
if [ "$WHAT" = asan ] || [ "$WHAT" = all ]; then
  log=profiles/r06_sanitize_asan_ubsan.log
  flags="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
  echo "== asan_ubsan: g++ $flags (tests/conftest.py build_emulator, PFV_EMU_DEFS); tools/sanitize_run.py -- source_hash $(python -c "import __graft_entry__ as g; print(g.source_hash())") -- $(date -u +%FT%TZ), $(gcc --version | head -1)" > $log
  PFV_EMU_DEFS="$flags" LD_PRELOAD="$GCC_LIBDIR/libasan.so" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 timeout 3000 python tools/sanitize_run.py >> $log 2>&1 || rc=1
  echo "== exit code $rc; sanitizer reports in this log: $(count $log)" >> $log; tail -2 $log
fi
if [ "$WHAT" = tsan ] || [ "$WHAT" = all ]; then
  # ThreadSanitizer, natively (no python: gcc 11's libtsan preloaded into python hangs at start-up in this container).  Two programs built with
  # -fsanitize=thread against the emulator library built the same way (the emulator's fibers are announced to the runtime, tests/hipemu/hipemu.cpp):
  #   tests/cpp/roundtrip.cpp       the C++ mirror's round trip, host parsers then device entropy stage forced
  #   tests/cpp/threads_driver.cpp  EVERY threaded object: pfv_decoder with 0 / 1 / 3 look-ahead threads, pfv_gop_decoder (parse pool of 3 + device
  #                                 windows + host fallback, two batch shapes), pfv_gop_encoder, the batch objects' pools -- host and device
  #                                 entropy, on the intact stream and on damaged ones; ONE PROCESS PER STREAM (the runtime's per-process limits on
  #                                 fibers / trace memory are reached after a few hundred emulated launches with threads coming and going)
  # (round 6 also tried /opt/rocm/llvm/bin/clang++ -fsanitize=thread with its static runtime, as the round-5 review suggested: it builds -- the
  # emulator needed PFV_WAVES_PER_EU and a __has_feature spelling for that -- but that runtime reports the scheduler <-> fiber hand-over of the
  # emulator's own globals as races and then dies inside its unwinder on the makecontext frames; gcc's runtime follows the same fibers cleanly.)
  log=profiles/r06_sanitize_tsan.log
  flags="-fsanitize=thread -fno-omit-frame-pointer"
  hash=$(python -c "import __graft_entry__ as g; print(g.source_hash())")
  echo "== tsan, native: g++ $flags -- emulator library + tests/cpp/roundtrip.cpp + tests/cpp/threads_driver.cpp -- source_hash $hash -- $(date -u +%FT%TZ), $(gcc --version | head -1)" > $log
  tmp=$(mktemp -d)
  exe=$(PFV_EMU_DEFS="$flags" python tools/sanitize_native.py $tmp 2>> $log | tail -1)
  if [ -x "$exe" ]; then
    for mode in "" 2; do
      echo "== roundtrip: PFV_TEST_ENTROPY_DECODE='$mode'" >> $log
      ( cd $tmp && PFV_TEST_ENTROPY_DECODE=$mode TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1" timeout 1500 $exe 64 48 30 5 4 -1 in.yuv out.pfv out.yuv ) >> $log 2>&1 || { rc=1; echo "== run failed" >> $log; }
    done
    N_DAMAGED=${PFV_TSAN_DAMAGED:-4}
    for si in $(seq 0 $N_DAMAGED); do
      echo "== threads_driver: stream $si of 0..$N_DAMAGED (0 = intact)" >> $log
      ( cd $tmp && TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1" timeout 2400 $tmp/threads_driver_san $N_DAMAGED $si ) >> $log 2>&1 || { rc=1; echo "== run failed" >> $log; }
    done
  else
    rc=1; echo "== build failed" >> $log
  fi
  rm -rf $tmp
  echo "== exit code $rc; sanitizer reports in this log: $(count $log)" >> $log; tail -2 $log
fi
exit $rc
