#!/bin/bash
# The GPU parity suite on the tests' wave emulator (tests/emu) under the host sanitizers -- device code included.  No GPU needed.
#   bash tools/sanitize_emu.sh asan    AddressSanitizer + UBSan: every kernel's loads and stores against the hipMalloc'ed blocks / the LDS it asked for
#   bash tools/sanitize_emu.sh tsan    ThreadSanitizer: the host side's threads with the kernels underneath; the resident trace server as a thread
#   bash tools/sanitize_emu.sh order   plain build, the lanes of a wave run in random order between cross-lane operations (dependences on lockstep)
# -> scratch/emu/<mode>_suite.txt; a summary on stdout.  (profiles/r5_sanitizers_emulator.txt holds round 5's.)
cd "$(dirname "$0")/.." || exit 1
mode=${1:-asan}; mkdir -p scratch/emu
EXCL="not rccl and not loaded_native and not full_size and not c5_ten and not c2_full_frame and not million_triangle and not c3_teapot and not bench and not literal_sampler"
export MALLIE_ALLOW_EMULATOR=1 MGPU_TRACE_SERVER_TIMEOUT_MS=1800000
case $mode in
asan) read lib rt < <(python tests/emu/build_emu.py asan)
  LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_stack_use_after_return=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 MALLIE_MGPU_LIB=$lib \
    python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 3000 -k "$EXCL" --tb=line -p no:cacheprovider -rA > scratch/emu/asan_suite.txt 2>&1
  echo "sanitizer reports: $(grep -c 'AddressSanitizer\|runtime error:' scratch/emu/asan_suite.txt)"; grep -E "passed|failed" scratch/emu/asan_suite.txt | tail -1
  grep -qE "[0-9]+ passed" scratch/emu/asan_suite.txt || echo "NO SUMMARY LINE: the run died (a sanitizer abort takes pytest's buffered output with it) -- rerun the last test printed with -x -s" ;;
tsan) read lib rt < <(python tests/emu/build_emu.py tsan)
  LD_PRELOAD=$rt TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:history_size=4" MALLIE_NO_TORCH=1 MALLIE_MGPU_LIB=$lib \
    python -m pytest tests/emu/cases_emu.py -q -s -p no:cacheprovider --tb=short -k "resident_server or several_ranks" > scratch/emu/tsan_suite.txt 2>&1
  echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' scratch/emu/tsan_suite.txt)"; grep -E "passed|failed" scratch/emu/tsan_suite.txt | tail -1 ;;
order) lib=$(python tests/emu/build_emu.py)
  MGPU_EMU_LANE_ORDER=random MALLIE_MGPU_LIB=$lib python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 2400 -k "$EXCL" --tb=line -p no:cacheprovider -rA > scratch/emu/order_suite.txt 2>&1
  grep -E "passed|failed" scratch/emu/order_suite.txt | tail -1 ;;
esac
