#!/bin/bash
# ON THE GPU BOX: the adversarial leaf-hint test against the product library and against the round-4 library (must fail there),
# then C2 timing of every mallie_amd/ab/*.so.   usage: bash tools/r5_hint_check.sh
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5hint
python -m pytest tests -m gpu -x -q -k "leaf_hints or soups" > gpurun_out/r5hint/tests_product.txt 2>&1; echo "product rc=$?"
tail -3 gpurun_out/r5hint/tests_product.txt
if [ -f mallie_amd/ab/r4.so ]; then
  MALLIE_MGPU_LIB=mallie_amd/ab/r4.so python -m pytest tests -m gpu -x -q -k "det_threshold" > gpurun_out/r5hint/tests_r4lib.txt 2>&1; echo "round-4 library rc=$? (expected: 1)"
  grep -E "AssertionError|changed|passed|failed" gpurun_out/r5hint/tests_r4lib.txt | tail -4
fi
bash tools/ab_run.sh c2 2>&1 | tee gpurun_out/r5hint/ab_c2.txt
