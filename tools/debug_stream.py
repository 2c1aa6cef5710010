import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests")); os.chdir(_R)
import numpy as np
import mallie_amd as M
import oracle_lib as O
g = O.load_golden("cornell_obj")
sc = M.Scene(g["verts"], g["faces"], g["matIDs"], g["normals"], None)
plane = sc.plane()
W, H = 512, 512
frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
prev = None; state = None
for k in range(5):
    img, _, st, state, _ = sc.render_stream(frame, W, H, 16, 1, plane, stream_state=state)
    c = sc.stream_classes(W, H)
    if prev is not None:
        ys, xs = np.nonzero(c != prev)
        print("call %d: %s promoted:" % (k, sc.stream_stats()), list(zip(xs.tolist(), ys.tolist(), prev[ys, xs].tolist()))[:40])
    else:
        ys, xs = np.nonzero(c == 2)
        print("call 0: uncertain %d; rows histogram:" % len(ys), np.bincount(ys, minlength=H).nonzero()[0][:60], "...")
        print("   row 255/256/257 uncertain counts", (c[255] == 2).sum(), (c[256] == 2).sum(), (c[257] == 2).sum(), (c[258] == 2).sum())
    prev = c
