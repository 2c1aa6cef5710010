import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import mallie_amd as M
from mallie_amd import workloads
cfg = workloads.CONFIGS["c2"]
sc = workloads.make_scene(cfg)
W, H = 1920, 1080
frame = workloads.camera(cfg)
plane = sc.plane()
img = np.zeros((H, W, 3), "<f4"); cnt = np.zeros((H, W), "<i4")
for passes in (1, 4, 16):
    ts = []; ks = []
    for k in range(8):
        t0 = time.perf_counter()
        _, _, st = sc.render(frame, W, H, 5, passes, plane, M.RNG_HASH, seed=1, pass_base=k * passes, image=img, count=cnt)
        ts.append(1e3 * (time.perf_counter() - t0)); ks.append(st["kernel_ms"])
    sc.set_render_ahead(True)
    ta = []
    for k in range(12):
        t0 = time.perf_counter()
        sc.render(frame, W, H, 5, passes, plane, M.RNG_HASH, seed=1, pass_base=100 + k * passes, image=img, count=cnt, want_stats=False)
        ta.append(1e3 * (time.perf_counter() - t0))
    sc.set_render_ahead(False)
    print("mgpu_render 1080p, %2d pass(es) per call: %.2f ms per call (kernel %.2f ms); with the render-ahead (what mallie::Render uses): %.2f ms per call, %s" % (
        passes, np.median(ts[2:]), np.median(ks[2:]), np.median(ta[2:]), sc.render_ahead_stats()))
