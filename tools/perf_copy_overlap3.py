"""ON THE GPU BOX: ONE render stream (frames in order), copies on a second stream.  V3: the copy stream waits for the frame's event on
the GPU; V4: the host waits for the event, then enqueues the copy."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
from mallie_amd import workloads
cfg = workloads.CONFIGS["c2"]
sc = workloads.make_scene(cfg)
W, H, mpl, spp = 1920, 1080, 5, 16
cam = workloads.camera(cfg); plane = sc.plane()
dev = torch.device("cuda", 0)
imgs = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)]
hosts = [torch.empty((H, W, 3), dtype=torch.float32).pin_memory() for _ in range(2)]
s_r, s_c = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def render(k, i):
    sc.render_strips_device(cam, W, H, imgs[i].data_ptr(), H, maxPathLength=mpl, passes=spp, plane=plane, seed=1, pass_base=k * spp, stream=s_r.cuda_stream)
for mode in ("V3 one render stream, copy stream waits on the GPU", "V4 one render stream, host waits then enqueues", "V0 one render stream, no copies"):
    for k in range(3): render(k, k % 2)
    torch.cuda.synchronize()
    N = 20
    done = [None, None]; copied = [None, None]; durs = []
    t0 = time.perf_counter()
    for k in range(N):
        i = k % 2
        if copied[i] is not None: s_r.wait_event(copied[i])   # the slot's image is free when its copy has left
        render(k, i)
        done[i] = torch.cuda.Event(); done[i].record(s_r)
        if mode.startswith("V3"):
            s_c.wait_event(done[i])
            ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s_c):
                ca.record(); hosts[i].copy_(imgs[i], non_blocking=True); cb.record()
            copied[i] = cb; durs.append((ca, cb))
            if copied[1 - i] is not None: copied[1 - i].synchronize()   # the caller takes frame k - 1
        elif mode.startswith("V4") and k > 0:
            j = 1 - i
            done[j].synchronize()
            ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s_c):
                ca.record(); hosts[j].copy_(imgs[j], non_blocking=True); cb.record()
            copied[j] = cb; durs.append((ca, cb))
            cb.synchronize()
    torch.cuda.synchronize()
    print("%s: %.3f ms per frame; copy durations %s" % (mode, 1e3 * (time.perf_counter() - t0) / N, np.round([a.elapsed_time(b) for a, b in durs], 2)[:10]))
