import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import mallie_amd as M
import oracle_lib as O
g = O.load_golden("cornell_obj")
sc = M.Scene(g["verts"], g["faces"], g["matIDs"], g["normals"], None, g["nodes"], g["indices"])
osc = O.scene_from_golden("cornell_obj")
W, H = 96, 80
frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
for mpl in (2, 3, 16):
  for plane in (None, osc.plane()):
    img, count, st = sc.render(frame, W, H, mpl, 1, plane, M.RNG_HASH, seed=42)
    oimg, ocount, ost, _ = osc.render(frame, W, H, mpl, 1, plane, O.RNG_HASH, seed=42)
    bad = np.argwhere((img != oimg).any(-1))
    print("mpl", mpl, "plane", plane is not None, "bad", len(bad), st["real_rays"], ost["real_rays"], st["nodes"], ost["nodes"], st["tris"], ost["tris"], st["trace_calls"], ost["trace_calls"])
    for (y, x) in bad[:12]:
        print("   ", y, x, img[y, x, 0], oimg[y, x, 0])
