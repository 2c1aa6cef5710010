"""Experiment (round 4): what does the walk cost over a tree with smaller leaves than BVHBuildOptions' minLeafPrimitives = 16?
Same builder, same node / index format, same kernels: only the (nodes, indices) pair handed to mgpu_scene_create changes.
Prints the kernel time per frame, the work per ray and how many pixels differ from the frame over the reference's tree
(an exact tie between two triangles is resolved by visiting order, which is the tree's)."""
import sys, os, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, _os.path.join(_R, "tests")); _os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
import oracle_lib as O
from mallie_amd.scenes import suzanne_grid
which = sys.argv[1] if len(sys.argv) > 1 else "cornell"
leaves = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 8, 6, 4, 3, 2]
W, H = 1920, 1080
if which in ("cornell", "teapot"):
    g = O.load_golden(which + "_obj")
    verts, faces, mats, normals = g["verts"].astype(np.float64), g["faces"], g["matIDs"], (g["normals"] if g["has_normals"] else None)
    eye, la, mpl, spp = ((0, 0, 20), (0, 0, 0), 5, 16) if which == "cornell" else ((0, 40, 250), (0, 40, 0), 9, 64)
else:
    n = int(which[4:])
    c = O.load_golden("cornell_obj")
    verts, faces, mats, normals = suzanne_grid(c["verts"], c["faces"], n)
    eye, la, mpl, spp = (0, 40, 80), (0, 0, 0), 5, 16
    if n > 64: W, H, spp = 3840, 2160, 64
spp = int(os.environ.get("SPP", spp))
frame = M.camera_frame(eye, la, width=W, height=H)
ref = None
for ml in leaves:
    t = time.time(); nodes, idx, st = M.bvh_build(verts, faces, minLeaf=ml, device=0); tb = time.time() - t
    sc = M.Scene(verts, faces, mats, normals, None, nodes, idx)
    plane = sc.plane()
    buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    ts = []
    for i in range(4):
        buf.fill_(float("nan"))
        s = sc.render_strips_device(frame, W, H, buf.data_ptr(), H, maxPathLength=mpl, passes=spp, plane=plane, seed=1, want_stats=True)
        ts.append(s["kernel_ms"])
    img = buf.cpu().numpy()
    if ref is None: ref = img
    d = int((img != ref).any(-1).sum())
    print("%s minLeaf %2d: nodes %8d depth %3d build %.2fs | kernel %.2f ms  %.0f Mrays/s  rays %d nodes/ray %.2f tris/ray %.2f | pixels differing from the first tree's frame: %d" % (
        which, ml, len(nodes), st["maxTreeDepth"], tb, float(np.median(ts[1:])), s["real_rays"] / np.median(ts[1:]) / 1e3, s["real_rays"], s["nodes"] / s["real_rays"], s["tris"] / s["real_rays"], d), flush=True)
    del sc
