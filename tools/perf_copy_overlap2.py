"""ON THE GPU BOX: does a D2H copy that WAITS (on the GPU) for a kernel's event overlap the next kernel?  V1: stream-wait-event then
copy; V2: host waits for the event, then enqueues the copy."""
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
s_r = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
s_c = torch.cuda.Stream(dev)
def render(k, i):
    sc.render_strips_device(cam, W, H, imgs[i].data_ptr(), H, maxPathLength=mpl, passes=spp, plane=plane, seed=1, pass_base=k * spp, stream=s_r[i].cuda_stream)
modes = ("V1 stream-wait-event", "V2 host waits, then enqueues", "V0 no copies")
if os.environ.get("ONLY"): modes = modes[1:2]
for mode in modes:
    for k in range(3): render(k, k % 2)
    torch.cuda.synchronize()
    N = int(os.environ.get("FRAMES", 20))
    done = [torch.cuda.Event(), torch.cuda.Event()]; copied = [None, None]
    durs = []
    kev = []
    t0 = time.perf_counter()
    for k in range(N):
        i = k % 2
        if copied[i] is not None: s_r[i].wait_event(copied[i])
        ka = torch.cuda.Event(enable_timing=True); ka.record(s_r[i])
        render(k, i)
        done[i] = torch.cuda.Event(enable_timing=True); done[i].record(s_r[i])
        kev.append((ka, done[i]))
        if mode.startswith("V1"):
            s_c.wait_event(done[i])
            with torch.cuda.stream(s_c): hosts[i].copy_(imgs[i], non_blocking=True)
            copied[i] = torch.cuda.Event(); copied[i].record(s_c)
            if copied[1 - i] is not None: copied[1 - i].synchronize()   # the caller takes frame k - 1
        elif mode.startswith("V2"):
            if k > 0:
                j = 1 - i
                done[j].synchronize()
                ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(s_c):
                    ca.record(); hosts[j].copy_(imgs[j], non_blocking=True); cb.record()
                copied[j] = cb
                copied[j].synchronize()
                durs.append(ca.elapsed_time(cb))
    torch.cuda.synchronize()
    starts = np.array([kev[0][0].elapsed_time(a) for a, b in kev]); ends = np.array([kev[0][0].elapsed_time(b) for a, b in kev])
    print("   stream start->end per frame (ms): %s" % np.round(ends - starts, 2)[:12])
    print("   end-to-end spacing of consecutive frames (ms): %s" % np.round(np.diff(ends), 2)[:12])
    print("%s: %.3f ms per frame" % (mode, 1e3 * (time.perf_counter() - t0) / N), ("copy durations (HIP events) %s" % np.round(durs, 2)) if durs else "")
