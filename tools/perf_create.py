"""Scene set-up cost on the GPU box: mesh -> BVH build (device/host) -> mgpu_scene_create -> first render."""
import sys, os, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, _os.path.join(_R, "tests")); _os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
import oracle_lib as O
from mallie_amd.scenes import suzanne_grid
for n in [int(a) for a in sys.argv[1:]] or [32]:
    c = O.load_golden("cornell_obj")
    verts, faces, mats, normals = suzanne_grid(c["verts"], c["faces"], n)
    t = time.time(); nodes, idx, r = M.bvh_build(verts, faces, device=0); tb = time.time() - t
    for rep in range(2):
        t = time.time(); sc = M.Scene(verts, faces, mats, normals, None, nodes, idx); torch.cuda.synchronize(); tc = time.time() - t
        print("grid%d tris %d: device build %.3fs (kernels %.1f ms)  scene_create %.3fs  device MB %.0f" % (n, len(faces), tb, r["device_ms"], tc, sc.device_bytes() / 1e6), flush=True)
        if rep == 0: del sc
