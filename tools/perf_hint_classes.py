"""ON THE GPU BOX (diagnostic -DMGPU_UTIL -DMGPU_UTIL_HINTCLASS build in MALLIE_MGPU_LIB): leaf-hint consultations of a C2 frame by
ray class -- primary rays, bounce rays by the distance of their origin from the leaf -- and the triangle tests each class drops."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
g = np.load("tests/golden/cornell_obj.npz")
sc = M.Scene(g["verts"].astype(np.float64), g["faces"], g["matIDs"], g["normals"], None)
W, H, mpl, spp = 1920, 1080, 5, 16
frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
st = sc.render_strips_device(frame, W, H, buf.data_ptr(), H, maxPathLength=mpl, passes=spp, plane=sc.plane(), seed=1, want_stats=True)
w = [int(x) for x in sc.debug_words()]
names = ["primary", "bounce, origin < 1.5 from the leaf", "bounce, 1.5 .. 5", "bounce, > 5"]
cons = [x & 0xffffffff for x in w[28:32]]; drop = [x >> 32 for x in w[28:32]]
tot_c = sum(cons); tot_d = sum(drop)
print("%s: kernel %.2f ms, rays %d, tris/ray %.2f" % (os.environ.get("MALLIE_MGPU_LIB", "default"), st["kernel_ms"], st["real_rays"], st["tris"] / st["real_rays"]))
for k in range(4):
    print("  %-36s consultations %9d (%5.1f%%)  dropped tests %10d (%5.1f%%)  per consultation %.2f" % (
        names[k], cons[k], 100.0 * cons[k] / max(1, tot_c), drop[k], 100.0 * drop[k] / max(1, tot_d), drop[k] / max(1, cons[k])))
