#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 PMC passes (no tracing in the same run) over tools/perf_scenes.py <scene>.
# usage: bash tools/pmc_scene.sh <tag> <scene> "<counters pass 1>" "<counters pass 2>" ...
tag=$1; scene=$2; shift 2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p "$out"
i=0
for set in "$@"; do
  i=$((i+1))
  SPP=${SPP:-16} timeout 600 rocprofv3 --pmc $set --output-format csv -d "$out/pmc$i" -o p -- python tools/perf_scenes.py $scene > "$out/pmc$i.log" 2>&1
  python tools/pmc_sum.py "$out/pmc$i" >> "$out/summary.txt" 2>&1
done
cat "$out/summary.txt"
