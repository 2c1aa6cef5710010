import sys, os, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, _os.path.join(_R, "tests")); _os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
import oracle_lib as O
from mallie_amd.scenes import suzanne_grid
which = sys.argv[1] if len(sys.argv) > 1 else "grid32"
W, H = 1920, 1080
if which == "teapot":
    g = O.load_golden("teapot_obj")
    verts, faces, mats, normals = g["verts"].astype(np.float64), g["faces"], g["matIDs"], g["normals"]
    eye, la, mpl, spp = (0, 40, 250), (0, 40, 0), 9, 64
else:
    n = int(which[4:])
    c = O.load_golden("cornell_obj")
    verts, faces, mats, normals = suzanne_grid(c["verts"], c["faces"], n)
    eye, la, mpl, spp = (0, 40, 80), (0, 0, 0), 5, 16
    if n > 64: W, H, spp = 3840, 2160, 64
t = time.time(); nodes, idx, st = M.bvh_build(verts, faces); tb = time.time() - t
sc = M.Scene(verts, faces, mats, normals, None, nodes, idx)
print(which, "tris", len(faces), "nodes", len(nodes), "depth", st["maxTreeDepth"], "build %.2fs" % tb, "device MB %.1f" % (sc.device_bytes() / 1e6))
frame = M.camera_frame(eye, la, width=W, height=H)
plane = sc.plane()
buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
spp_run = int(os.environ.get("SPP", spp))
ts = []
for i in range(3):
    buf.fill_(float("nan"))  # a pixel the kernel skips must not hide behind the previous frame
    s = sc.render_strips_device(frame, W, H, buf.data_ptr(), H, maxPathLength=mpl, passes=spp_run, plane=plane, seed=1, want_stats=True)
    ts.append(s["kernel_ms"])
ms = min(ts)
alg = s["nodes"] * 64 + s["tris"] * 76 + s["real_rays"] * 80
assert not bool(torch.isnan(buf).any().item()), "unrendered pixels"
print("  %dx%d spp %d mpl %d: kernel %.2f ms  %.0f Mrays/s  rays %d nodes/ray %.2f tris/ray %.2f  alg %.2f GB -> %.0f GB/s" % (
    W, H, spp_run, mpl, ms, s["real_rays"] / ms / 1e3, s["real_rays"], s["nodes"] / s["real_rays"], s["tris"] / s["real_rays"], alg / 1e9, alg / ms / 1e6))
# parity spot check vs oracle on a few rows (oracle with the same BVH)
osc = O.OracleScene(verts, faces, mats, normals, None, nodes, idx)
y0 = H // 2 - 2
oimg, _, ost, _ = osc.render(frame, W, H, mpl, 2, plane, O.RNG_HASH, seed=1, window=(0, y0, W, y0 + 4))
img2 = torch.empty((4, W, 3), dtype=torch.float32, device="cuda")
sc.render_strips_device(frame, W, H, img2.data_ptr(), 4, y_first=y0, strip_h=4, y_period=4, maxPathLength=mpl, passes=2, plane=plane, seed=1, want_stats=True)
a = img2.cpu().numpy(); b = oimg[y0:y0 + 4]
print("  parity rows %d..%d: bit-exact %s, differing pixels %d" % (y0, y0 + 3, a.tobytes() == b.tobytes(), int((a != b).any(-1).sum())))
