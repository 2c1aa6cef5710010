#!/bin/bash
# Scene::Trace calls/s from 1 / 4 / 16 / 64 host threads: resident server, submission queue, a launch per call.
# usage (on the GPU box): tools/perf_trace_calls.sh [n_rays]
set -e
cd "$(dirname "$0")/.."
N=${1:-4000}
T=$(mktemp -d)
g++ -O1 -std=c++11 -pthread -I include/mallie tests/cpp/facade_driver.cc -L mallie_amd -lmallie_mgpu -Wl,-rpath,$PWD/mallie_amd -Wl,-rpath,/opt/rocm/lib -o $T/drv
python - "$T" "$N" <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
import oracle_lib as O
T, n = sys.argv[1], int(sys.argv[2])
g = O.load_golden("cornell_obj")
with open(T + "/scene.obj", "w") as f:
    for v in g["verts"]: f.write("v %r %r %r\n" % (float(v[0]), float(v[1]), float(v[2])))
    for a, b, c in g["faces"]: f.write("f %d %d %d\n" % (a + 1, b + 1, c + 1))
t = O.load_golden("trace_cornell_obj")
r = t["rays"]
rays = np.tile(r, ((n + len(r) - 1) // len(r), 1))[:n]
np.ascontiguousarray(rays, "<f8").tofile(T + "/rays.bin")
PY
for mode in server queue launch; do
  for nt in 1 4 16 64; do
    case $mode in
      server) E="";;
      queue) E="MGPU_TRACE_SERVER=0";;
      launch) E="MGPU_TRACE_SERVER=0 MGPU_TRACE_QUEUE=0";;
    esac
    echo "== $mode, $nt threads"
    (cd $T && env $E timeout 120 ./drv trace_mt obj scene.obj rays.bin out_${mode}_$nt.bin $nt | grep trace_mt)
  done
done
cmp $T/out_server_1.bin $T/out_launch_1.bin && cmp $T/out_server_16.bin $T/out_launch_1.bin && cmp $T/out_server_64.bin $T/out_queue_4.bin && echo "records identical"
