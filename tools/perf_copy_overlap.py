"""ON THE GPU BOX: a 24.9 MB device-to-host copy alone, and under a running C2 render kernel (and what it does to the kernel)."""
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
img = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
src = torch.rand((H, W, 3), dtype=torch.float32, device=dev)
host = torch.empty((H, W, 3), dtype=torch.float32).pin_memory()
s_r, s_c = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def ev(): return torch.cuda.Event(enable_timing=True)
def render(stream):
    sc.render_strips_device(cam, W, H, img.data_ptr(), H, maxPathLength=mpl, passes=spp, plane=plane, seed=1, stream=stream.cuda_stream)
for _ in range(3): render(s_r)
torch.cuda.synchronize()
# copy alone
ts = []
for _ in range(5):
    a, b = ev(), ev()
    with torch.cuda.stream(s_c):
        a.record(); host.copy_(src, non_blocking=True); b.record()
    torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print("copy alone: %.3f ms (%.1f GB/s)" % (np.median(ts), 24.8832 / np.median(ts)))
# render alone
ts = []
for _ in range(5):
    a, b = ev(), ev()
    with torch.cuda.stream(s_r):
        a.record(); render(s_r); b.record()
    torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print("render alone: %.3f ms" % np.median(ts))
# copy under render: start the copy 1 ms after the render was enqueued
tc, tr = [], []
for _ in range(5):
    a, b, c, d = ev(), ev(), ev(), ev()
    with torch.cuda.stream(s_r):
        a.record(); render(s_r); b.record()
    time.sleep(0.001)
    with torch.cuda.stream(s_c):
        c.record(); host.copy_(src, non_blocking=True); d.record()
    torch.cuda.synchronize(); tr.append(a.elapsed_time(b)); tc.append(c.elapsed_time(d))
    print("   copy started %.2f ms after the render, ended %.2f ms after it started" % (a.elapsed_time(c), a.elapsed_time(d)))
print("copy under render: copy %.3f ms, render %.3f ms" % (np.median(tc), np.median(tr)))
