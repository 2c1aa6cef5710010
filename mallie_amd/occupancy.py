"""Active-lane accounting of the render kernel on one BASELINE configuration: python -m mallie_amd.occupancy c2 [frames]

Meant to run with MALLIE_MGPU_LIB pointing at libmallie_mgpu_occ.so (the same sources built with -DMGPU_OCC=1: the
accounting costs 3-4 % in registers, so the product library is built without it).  Prints one JSON line."""
import json
import sys

import torch

from . import mgpu, workloads


def main():
    cfg = workloads.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sc = workloads.make_scene(cfg)
    W, H = cfg["width"], cfg["height"]
    frame = workloads.camera(cfg)
    buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    plane = sc.plane() if cfg["plane"] else None
    sc.stats_read(reset=True)
    for i in range(frames):
        sc.render_strips_device(frame, W, H, buf.data_ptr(), H, maxPathLength=cfg["bounces"] + 1, passes=cfg["spp"], plane=plane,
                                seed=cfg["seed"], pass_base=i * cfg["spp"])
    o = sc.occupancy()
    o["library"] = mgpu.lib_path()
    print(json.dumps(o))


if __name__ == "__main__":
    main()
