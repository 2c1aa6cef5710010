"""Throughput of rank 0's share of the C2 frame with 1 or 2 frames in flight (two streams), measured on ONE GPU.
Experiment behind DESIGN.md 6: does the end-of-launch tail of frame k fill with frame k+1's work?"""
import sys, os, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); sys.path.insert(0, _R); _os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
from mallie_amd.frame import strip_rows
g = np.load("tests/golden/cornell_obj.npz")
W, H, mpl, spp = 1920, 1080, 5, 16
frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
NS = int(os.environ.get("NSLOTS", "2"))
sc = M.Scene(g["verts"].astype(np.float64), g["faces"], g["matIDs"], g["normals"], None)
scs = [sc] * NS  # one scene: the library keeps one launch scratch set per stream
plane = sc.plane()
streams = [torch.cuda.Stream() for _ in range(NS)]
for world in (1, 2, 4, 8):
    rank = 0
    n_rows = len(strip_rows(H, world, rank))
    bufs = [torch.full((n_rows, W, 3), float("nan"), dtype=torch.float32, device="cuda") for _ in range(NS)]
    def go(i, slots):
        k = i % slots
        scs[k].render_strips_device(frame, W, H, bufs[k].data_ptr(), n_rows, y_first=rank * 8, strip_h=8, y_period=8 * world,
                                    maxPathLength=mpl, passes=spp, plane=plane, seed=1, stream=streams[k].cuda_stream)
    out = []
    for slots in range(1, NS + 1):
        for i in range(6): go(i, slots)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 40
        for i in range(K): go(i, slots)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / K * 1e3)
    same = all(torch.equal(bufs[0], b) for b in bufs[1:])
    print("world %d: rank0 " % world + "  ".join("%d in flight %.3f ms/frame" % (i + 1, t) for i, t in enumerate(out)) +
          "  (ideal %.3f)  frames equal: %s" % (out[0] if world == 1 else 0, same), flush=True)
