#!/bin/bash
# ON THE GPU BOX: the N > 1 machinery with N ranks sharing the one GPU (MGPU_FRAME_TRANSPORT=copy): what a render call costs on the
# host (ONE thread enqueues every member's work) and what the frames cost, one frame per launch (the latency case) and eight.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r4multi}; mkdir -p $out
for n in 2 4 8; do for fpl in 1 8; do
  MGPU_FRAME_TRANSPORT=copy python bench.py --gpus $n --frames-per-launch $fpl --steps 24 --warmup 8 --no-extras --no-cpu-baseline > $out/n${n}_fpl$fpl.json 2> $out/n${n}_fpl$fpl.err
  python - $out/n${n}_fpl$fpl.json $n $fpl <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print("ranks %s on one GPU, %s frame(s) per launch: %.3f ms per frame, %.0f Mrays/s; enqueue %.4f ms per render call; exchange %.4f ms per frame (%s, %s); kernel ms per launch by rank %s; frame equals single-GPU frame: %s" % (
    sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], c.get("enqueue_ms_per_call") or -1, c.get("exchange_ms_per_frame") or -1, c.get("transport"), c.get("exchange_mode"),
    c["kernel_ms_per_launch_by_rank"], c.get("frame_equals_single_gpu_frame")))
PY
done; done
