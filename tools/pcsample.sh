#!/bin/bash
# ON THE GPU BOX: PC sampling of the C2 render kernel (rocprofv3 beta feature).  usage: bash tools/pcsample.sh <tag> [method] [unit] [interval]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-pcs}; method=${2:-host_trap}; unit=${3:-time}; interval=${4:-1}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $method --pc-sampling-interval $interval \
  --output-format csv -d $out/raw -- python $GRAFT_REPO_ROOT/tools/perf_c2.py > $out/log.txt 2>&1
echo "rc=$?"; tail -5 $out/log.txt; find $out/raw -type f | head; 
f=$(find $out/raw -name "*pc_sampling*csv" | head -1)
if [ -n "$f" ]; then wc -l $f; head -3 $f; python3 - "$f" "$out/top.txt" <<'PY'
import csv, sys, collections
c = collections.Counter(); n = 0
with open(sys.argv[1]) as fh:
    r = csv.DictReader(fh)
    cols = r.fieldnames
    for row in r:
        n += 1
        c[(row.get("Instruction") or row.get("Instruction_Comment") or "", row.get("Instruction_Comment") or "")] += 1
with open(sys.argv[2], "w") as o:
    o.write("columns: %s\nsamples %d\n" % (cols, n))
    for k, v in c.most_common(400): o.write("%7d %5.2f%%  %s | %s\n" % (v, 100.0 * v / max(1, n), k[0], k[1]))
PY
head -40 $out/top.txt; rm -rf $out/raw; fi
