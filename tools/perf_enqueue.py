"""ON THE GPU BOX: host time of mgpu_frame_render_batch with N ranks sharing the GPU (copy transport), timing ring on / off."""
import sys, os, time
os.environ["MGPU_FRAME_TRANSPORT"] = "copy"
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
from mallie_amd import workloads
cfg = workloads.CONFIGS["c2"]
N = int(os.environ.get("RANKS", 8))
scenes = [workloads.make_scene(cfg) for _ in range(N)]
W, H, mpl, spp = 1920, 1080, 5, 16
cam = workloads.camera(cfg); plane = scenes[0].plane()
for fpl in (1, 8):
    for timing in (False, True):
        fr = M.Frame(scenes, [0] * N, W, H, strip_h=8, frames_in_flight=16)
        for sc in scenes: sc.timing_enable(timing)
        k = 0
        def batch():
            global k
            slots = fr.render_batch(cam, mpl, spp, fpl, plane, seed=1, pass_base=k * spp) if fpl > 1 else [fr.render(cam, mpl, spp, plane, seed=1, pass_base=k * spp)]
            k += fpl
            return slots
        for _ in range(3): last = batch()
        for s in last: fr.wait(s)
        torch.cuda.synchronize()
        hosts = []
        t0 = time.perf_counter()
        nb = 48 // fpl
        for _ in range(nb):
            a = time.perf_counter(); last = batch(); hosts.append(1e3 * (time.perf_counter() - a))
        for s in last: fr.wait(s)
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0) / (nb * fpl)
        for sc in scenes:
            if timing: sc.timing_read()
            sc.timing_enable(False)
        print("%d ranks, %d frame(s) per launch, timing ring %-5s: %.3f ms per frame; host ms per render call: median %.3f max %.3f" % (N, fpl, timing, dt, np.median(hosts), max(hosts)))
        fr.close()
