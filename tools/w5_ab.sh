#!/bin/bash
# ON THE GPU BOX: k_render_w5 (MGPU_W5=1) beside k_render_sm on C3 / C4 / C5: kernel ms, rays and the frame's checksum (must be equal).
cd "$GRAFT_REPO_ROOT" || exit 1
cfgs=${@:-c4 c3 c5}
for c in $cfgs; do
  n=8; [ $c = c5 ] && n=4
  timeout 600 python tools/perf_cfg.py $c $n 2>&1 | tail -1
  MGPU_W5=1 timeout 600 python tools/perf_cfg.py $c $n 2>&1 | tail -1
done
