#!/bin/bash
# ON THE GPU BOX: dynamic instruction mix of k_render_sm (rocprofv3 --pmc, two passes per workload).  usage: tools/pmc_mix.sh <tag> [workloads...]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=${1:-mix}; shift
out=gpurun_out/$tag; mkdir -p $out
for w in ${@:-c2 c4}; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --kernel-include-regex "k_render_sm" --output-format csv -d $out/${w}_mix1 -o p -- python tools/pmc_workload.py $w 3 > $out/${w}_mix1.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-include-regex "k_render_sm" --output-format csv -d $out/${w}_mix2 -o p -- python tools/pmc_workload.py $w 3 > $out/${w}_mix2.log 2>&1
done
python - "$out" "$@" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for w in (sys.argv[2:] or ["c2", "c4"]):
    acc = collections.defaultdict(float)
    for f in glob.glob("%s/%s_mix*/**/*counter_collection.csv" % (out, w), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]) / 3
    print(w, "per frame:", ", ".join("%s %.3e" % (k, v) for k, v in sorted(acc.items())))
PY
