import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
np.set_printoptions(precision=17, linewidth=200)
import mallie_amd as M
import oracle_lib as O
g = O.load_golden("cornell_obj")
sc = M.Scene(g["verts"], g["faces"], g["matIDs"], g["normals"], None, g["nodes"], g["indices"])
osc = O.scene_from_golden("cornell_obj")
W, H = 96, 80
frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
plane = osc.plane()
mpl = 3
img, count, st = sc.render(frame, W, H, mpl, 1, plane, M.RNG_HASH, seed=42)
oimg, ocount, ost, _ = osc.render(frame, W, H, mpl, 1, plane, O.RNG_HASH, seed=42)
bad = np.argwhere((img != oimg).any(-1))
print("bad", len(bad))
names = "ox oy oz dx dy dz t hit slot nx ny nz mat L thr rad".split()
for (y, x) in bad[:3]:
    s0 = M.hash_state(42, 0, y * W + x)
    a = sc.probe_path(frame, W, H, x, y, s0, mpl, plane)
    b, rad = osc.probe_path(frame, x, y, s0, mpl, plane)
    print("pixel", y, x, "gpu iters", len(a), "oracle iters", len(b), "oracle rad", rad, "img", img[y, x, 0], oimg[y, x, 0])
    for i in range(max(len(a), len(b))):
        for k, nm in enumerate(names):
            va = a[i][k] if i < len(a) else None
            vb = b[i][k] if i < len(b) else None
            flag = "" if (va == vb or nm == "slot") else "   <<<<"
            print("  it%d %-4s gpu %-26r oracle %-26r%s" % (i, nm, va, vb, flag))
