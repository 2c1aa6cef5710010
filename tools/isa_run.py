#!/usr/bin/env python3
"""A BASELINE configuration at reduced resolution with its kernels executed from hipcc's gfx950 machine code by the tests' ISA interpreter
(tests/emu/isa_interp.cc; no GPU): dynamic instruction counts per ray by class -- the quantity rocprofv3's SQ_INSTS_VALU / SQ_INSTS_SALU
count on the hardware -- the frame's checksum, and optionally the per-instruction profile (tools/isa_profile.py reads it).
usage: MGPU_EMU_ISA=<hipcc -S dumps, ':'-separated> python tools/isa_run.py c2|c3|c4|c5 W H spp [profile.tsv]
       (tools/isa_profile.py builds the dumps and calls this)"""
import os, sys, time, ctypes, hashlib, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests"); os.chdir(R)
os.environ["MALLIE_ALLOW_EMULATOR"] = "1"; os.environ["MALLIE_NO_TORCH"] = "1"
os.environ.setdefault("MALLIE_MGPU_LIB", R + "/tests/emu/libmallie_mgpu_emu.so")
import numpy as np
import mallie_amd as M
from mallie_amd import workloads
L = ctypes.CDLL(os.environ["MALLIE_MGPU_LIB"])
cnt = (ctypes.c_ulonglong * 16).in_dll(L, "isa_counters")
key, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = dict(workloads.CONFIGS[key])
t = time.time()
sc = workloads.make_scene(cfg)
print("scene %.1f s" % (time.time() - t), flush=True)
cfg["width"], cfg["height"] = W, H
frame = workloads.camera(cfg)
L.isa_profile_reset()
t = time.time()
img, count, st = sc.render(frame, W, H, cfg["bounces"] + 1, spp, sc.plane() if cfg["plane"] else None, M.RNG_HASH, seed=cfg["seed"])
dt = time.time() - t
c = [int(x) for x in cnt]
rays = st["real_rays"]
print(json.dumps(dict(config=key, W=W, H=H, spp=spp, seconds=round(dt, 1), rays=rays, nodes_per_ray=round(st["nodes"] / rays, 3), tris_per_ray=round(st["tris"] / rays, 3),
                      isa_launches=c[8], valu=c[0], salu=c[1], branch=c[2], lds=c[3], vmem=c[4], valu_per_ray=round(c[0] / rays, 2), salu_per_valu=round(c[1] / max(c[0], 1), 3),
                      sha256=hashlib.sha256(img.tobytes()).hexdigest()[:16])))
if len(sys.argv) > 5:
    L.isa_profile_dump(sys.argv[5].encode())
