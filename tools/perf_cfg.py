#!/usr/bin/env python3
"""ON THE GPU BOX: kernel ms of `frames` consecutive frames of a BASELINE configuration (c2|c3|c4|c5) with the library in
MALLIE_MGPU_LIB (default: the product) and whatever MGPU_* switches the environment carries.  usage: python tools/perf_cfg.py c4 [frames]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); os.chdir(R)
import numpy as np, torch
from mallie_amd import workloads
cfg = workloads.CONFIGS[sys.argv[1]]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = workloads.make_scene(cfg)
W, H = cfg["width"], cfg["height"]
frame = workloads.camera(cfg)
buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
ts = []
for k in range(frames):
    st = sc.render_strips_device(frame, W, H, buf.data_ptr(), H, maxPathLength=cfg["bounces"] + 1, passes=cfg["spp"], plane=sc.plane(),
                                 seed=cfg["seed"], pass_base=k * cfg["spp"], want_stats=True)
    ts.append(st["kernel_ms"])
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("MGPU_") or k == "MALLIE_MGPU_LIB")
print("%s [%s]: kernel ms per frame %s -> median of the last %d: %.3f  (rays %d, checksum %.6f)" % (
    sys.argv[1], tag, " ".join("%.2f" % t for t in ts), max(1, frames - 2), float(np.median(ts[2:] or ts)), st["real_rays"], float(buf.double().sum().item())))
