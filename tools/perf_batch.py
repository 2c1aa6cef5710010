"""Rank 0's eighth of C2 through mgpu_render_frames_device: ms per frame for 1 / 2 / 4 / 8 frames per call."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
g = np.load("tests/golden/cornell_obj.npz")
sc = M.Scene(g["verts"].astype(np.float64), g["faces"], g["matIDs"], g["normals"], None)
W, H, mpl, spp = 1920, 1080, 5, 16
cam = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
plane = sc.plane()
for world in (8, 4, 2, 1):
    rows = M.frame_rows(H, 8, world, 0)
    stream = torch.cuda.current_stream().cuda_stream
    line = "world %d:" % world
    for n in (1, 2, 4, 8):
        bufs = [torch.empty((rows, W, 3), dtype=torch.float32, device="cuda") for _ in range(n)]
        ptrs = [b.data_ptr() for b in bufs]
        base = [0]
        def go():
            sc.render_frames_device(cam, W, H, ptrs, rows, y_first=0, strip_h=8, y_period=8 * world, maxPathLength=mpl, passes=spp,
                                    plane=plane, seed=1, pass_base=base[0], stream=stream)
            base[0] += n * spp
        for _ in range(3): go()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = max(2, 24 // n)
        for _ in range(K): go()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / (K * n) * 1e3
        line += "  %d/launch %.3f ms" % (n, ms)
    print(line, flush=True)
