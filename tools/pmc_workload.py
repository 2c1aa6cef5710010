#!/usr/bin/env python3
"""Renders `frames` frames of a BASELINE configuration (c2|c3|c4|c5) with the product library -- the command the rocprofv3
passes of profiles/collect_pmc.sh wrap.  usage: python tools/pmc_workload.py c2 [frames]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); os.chdir(R)
import torch
from mallie_amd import workloads
cfg = workloads.CONFIGS[sys.argv[1]]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc = workloads.make_scene(cfg)
W, H = cfg["width"], cfg["height"]
frame = workloads.camera(cfg)
buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for k in range(frames):
    st = sc.render_strips_device(frame, W, H, buf.data_ptr(), H, maxPathLength=cfg["bounces"] + 1, passes=cfg["spp"], plane=sc.plane(),
                                 seed=cfg["seed"], pass_base=k * cfg["spp"], want_stats=True)
print(sys.argv[1], "kernel_ms", st["kernel_ms"], "rays", st["real_rays"])
