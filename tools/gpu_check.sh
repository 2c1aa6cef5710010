#!/bin/bash
# ON THE GPU BOX: the GPU test-suite (or a -k subset) with its summary where gpurun's tail shows it.
# usage: bash tools/gpu_check.sh <tag> [pytest -k expression]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-check}; shift
mkdir -p gpurun_out/$tag
if [ -n "$1" ]; then python -m pytest tests -m gpu -x -q -k "$1" > gpurun_out/$tag/tests.txt 2>&1; else python -m pytest tests -m gpu -x -q > gpurun_out/$tag/tests.txt 2>&1; fi
echo "pytest rc=$?"
grep -E "passed|failed|error|parity:" gpurun_out/$tag/tests.txt | tail -6
