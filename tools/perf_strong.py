"""Per-GPU share of the C2 frame under strong scaling, measured on ONE GPU: rank 0's strips of a `world`-way split."""
import sys, os, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); sys.path.insert(0, _R); _os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
from mallie_amd.frame import strip_rows
g = np.load("tests/golden/cornell_obj.npz")
sc = M.Scene(g["verts"].astype(np.float64), g["faces"], g["matIDs"], g["normals"], None)
W, H, mpl, spp = 1920, 1080, 5, 16
frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
plane = sc.plane()
for world in (1, 2, 4, 8):
    res = []
    for rank in range(world):
        n_rows = len(strip_rows(H, world, rank))
        buf = torch.empty((n_rows, W, 3), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        def go():
            sc.render_strips_device(frame, W, H, buf.data_ptr(), n_rows, y_first=rank * 8, strip_h=8, y_period=8 * world,
                                    maxPathLength=mpl, passes=spp, plane=plane, seed=1, stream=stream)
        for _ in range(3): go()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 20
        for _ in range(K): go()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / K * 1e3)
        if rank >= 1: break
    print("world %d: rank0 %.3f ms/frame%s  (ideal %.3f)" % (world, res[0], "" if len(res) < 2 else ", rank1 %.3f" % res[1], 9.3 / world), flush=True)
