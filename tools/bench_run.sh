#!/bin/bash
# ON THE GPU BOX: bench.py (extra args passed on) into gpurun_out/<tag>/bench.json + a short summary.  usage: bash tools/bench_run.sh <tag> [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-bench}; shift
mkdir -p gpurun_out/$tag
python bench.py "$@" > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
echo "bench rc=$?"; tail -c 300 gpurun_out/$tag/bench.err | grep -v amdgpu.ids
python - gpurun_out/$tag/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, "kernel", d["roofline"].get("kernel_avg_ms"), "frac", d["roofline"].get("frac"))
print("readback:", d["config"].get("readback", "")[:60], d["config"].get("host_frame_equals_rerendered_frame"))
for k in ("frame_with_synchronous_readback", "frame_resident_in_hbm", "tile_order_off"):
    if k in d: print(k, d[k].get("ms_per_frame"))
if "reference_stream_1080p" in d: print("stream:", {k: v for k, v in d["reference_stream_1080p"].items() if k != "note"})
if "roofline" in d and d["roofline"].get("lane_occupancy"): print("occ:", d["roofline"]["lane_occupancy"])
for k, v in d.get("extra_configs", {}).items(): print(k, {x: v.get(x) for x in ("ms_per_frame", "kernel_ms_per_frame", "value", "error")}, (v.get("roofline") or {}).get("frac"))
if "cpu_baseline" in d: print("cpu:", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind", "gpu_frame_byte_equal")})
if "fast_mode_fp32" in d: print("fast:", d["fast_mode_fp32"].get("ms_per_frame"), d["fast_mode_fp32"].get("distance_to_fp64_frame", {}).get("rms_per_pixel_l2"))
PY
