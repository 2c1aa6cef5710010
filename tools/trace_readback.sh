#!/bin/bash
# ON THE GPU BOX: kernel + memory-copy timeline of the read-back pipeline (two frames in flight)
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4rb}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ONLY=2 FRAMES=8 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/raw -- python $GRAFT_REPO_ROOT/tools/${2:-perf_readback.py} > $out/log.txt 2>&1
tail -3 $out/log.txt
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40]))
for f in glob.glob(out + "/raw/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
t0 = rows[0][0]
with open(out + "/timeline.txt", "w") as o:
    for a, b, n in rows[-90:]:
        o.write("%10.3f %10.3f %8.3f  %s\n" % ((a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6, n))
print(open(out + "/timeline.txt").read()[-5000:])
PY
rm -rf $out/raw
