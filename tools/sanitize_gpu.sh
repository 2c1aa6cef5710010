#!/bin/bash
# SURVEY.md 5 / round-4 review item 7: the threaded host side of the library (mgpu_api.hip: mailbox server, submission queue,
# render-ahead, scene life cycle) under AddressSanitizer and ThreadSanitizer.
#   bash tools/sanitize_gpu.sh build    (anywhere: hipcc cross-compiles)  -> mallie_amd/ab/{asan,tsan}.so + tests/cpp/stress_{plain,asan,tsan}
#   bash tools/sanitize_gpu.sh run      (ON THE GPU BOX)                   -> gpurun_out/sanitize/*.txt, summary on stdout
# Host code only is instrumented (-fno-gpu-sanitize); the HIP runtime itself is not, so ThreadSanitizer runs with the suppressions
# below for races it reports INSIDE libamdhip64 / libhsa-runtime64.
cd "$(dirname "$0")/.." || exit 1
CLANG=/opt/rocm/lib/llvm/bin/clang++
RT=$(dirname $($CLANG -print-file-name=libclang_rt.asan-x86_64.so))
LIBDIR=$PWD/mallie_amd/ab
if [ "$1" = build ]; then
  python - <<'PY'
import os
from mallie_amd import build as b
d = os.path.join(os.path.dirname(b.LIB), "ab"); os.makedirs(d, exist_ok=True)
b.build_variant(os.path.join(d, "asan.so"), ["-fsanitize=address", "-fno-gpu-sanitize", "-shared-libasan", "-g", "-fno-omit-frame-pointer"])
b.build_variant(os.path.join(d, "tsan.so"), ["-fsanitize=thread", "-fno-gpu-sanitize", "-g", "-fno-omit-frame-pointer"])
PY
  g++ -O1 -std=c++11 -pthread tests/cpp/stress_driver.cc -L mallie_amd -lmallie_mgpu -Wl,-rpath,'$ORIGIN/../../mallie_amd' -Wl,-rpath,/opt/rocm/lib -o tests/cpp/stress_plain || exit 1
  $CLANG -O1 -g -std=c++11 -pthread -fsanitize=address -shared-libasan -fno-omit-frame-pointer tests/cpp/stress_driver.cc -l:asan.so -L $LIBDIR \
      -Wl,-rpath,'$ORIGIN/../../mallie_amd/ab' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$RT -o tests/cpp/stress_asan || exit 1
  $CLANG -O1 -g -std=c++11 -pthread -fsanitize=thread -fno-omit-frame-pointer tests/cpp/stress_driver.cc -l:tsan.so -L $LIBDIR \
      -Wl,-rpath,'$ORIGIN/../../mallie_amd/ab' -Wl,-rpath,/opt/rocm/lib -o tests/cpp/stress_tsan || exit 1
  ls -la tests/cpp/stress_* mallie_amd/ab/asan.so mallie_amd/ab/tsan.so
  exit 0
fi
out=gpurun_out/sanitize; mkdir -p $out
cat > $out/tsan.supp <<'SUP'
# the HIP / HSA runtimes are not instrumented: what ThreadSanitizer sees inside them is their business
called_from_lib:libamdhip64.so
called_from_lib:libhsa-runtime64.so
called_from_lib:libamd_comgr.so
race:libamdhip64.so
race:libhsa-runtime64.so
SUP
echo "== plain =="; timeout 300 tests/cpp/stress_plain > $out/plain.txt 2>&1; echo "rc=$?"; tail -1 $out/plain.txt
echo "== AddressSanitizer (host code of the library + driver) =="
STRESS_FAST_EXIT=1 ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:protect_shadow_gap=0 LD_LIBRARY_PATH=$RT:$LD_LIBRARY_PATH timeout 900 tests/cpp/stress_asan 16 600 10 > $out/asan.txt 2>&1; echo "rc=$?"
echo "AddressSanitizer reports: $(grep -c "ERROR: AddressSanitizer" $out/asan.txt)"; tail -2 $out/asan.txt
echo "== ThreadSanitizer (host code of the library + driver) =="
TSAN_OPTIONS="suppressions=$PWD/$out/tsan.supp:halt_on_error=0:second_deadlock_stack=1:history_size=4" timeout 900 tests/cpp/stress_tsan 16 600 10 > $out/tsan.txt 2>&1; echo "rc=$?"
echo "ThreadSanitizer reports: $(grep -c "WARNING: ThreadSanitizer" $out/tsan.txt)"; grep "WARNING: ThreadSanitizer" $out/tsan.txt | sort | uniq -c | head; tail -2 $out/tsan.txt
