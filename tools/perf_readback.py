"""ON THE GPU BOX: ms per C2 frame through the C ABI frame object, with / without the asynchronous read-back, 1..3 frames in flight."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
from mallie_amd import workloads
cfg = workloads.CONFIGS["c2"]
sc = workloads.make_scene(cfg)
W, H, mpl, spp = 1920, 1080, 5, 16
cam = workloads.camera(cfg); plane = sc.plane()
N = int(os.environ.get("FRAMES", 20))
only = os.environ.get("ONLY")
for fif in ((int(only),) if only else (1, 2, 3)):
    for rb in ((True,) if only else (False, True)):
        fr = M.Frame.create_rank(sc, 0, 0, 1, None, W, H, strip_h=8, frames_in_flight=fif)
        fr.set_readback(rb)
        def run(n, base):
            prev = []
            for k in range(n):
                s = fr.render(cam, mpl, spp, plane, seed=1, pass_base=(base + k) * spp)
                prev.append(s)
                if len(prev) >= fif:
                    q = prev.pop(0)
                    fr.wait_host(q) if rb else fr.wait(q)
            for q in prev:
                fr.wait_host(q) if rb else fr.wait(q)
        run(3, 0); torch.cuda.synchronize()
        sc.timing_enable(True)
        t0 = time.perf_counter(); run(N, 3); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        kms, nl = sc.timing_read(); sc.timing_enable(False)
        print("frames in flight %d, read-back %-5s: %.3f ms per frame (kernel events avg %.3f ms over %d launches)" % (fif, rb, 1e3 * dt / N, kms / max(nl, 1), nl))
        fr.close()
