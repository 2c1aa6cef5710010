"""k_render_env (RenderPanoramic) throughput on one GPU + the oracle on the host cores for the same frame."""
import sys, os, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, _os.path.join(_R, "tests")); _os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
import oracle_lib as O
g = O.load_golden("cornell_obj")
sc = M.Scene(g["verts"].astype(np.float64), g["faces"], g["matIDs"], g["normals"], None, g["nodes"], g["indices"])
osc = O.scene_from_golden("cornell_obj")
origin = np.array([0.0, 1.0, 4.0])  # main_console.cc:104-106, frame 0
for (W, H, stereo) in [(2048, 1024, 1), (4096, 2048, 1), (2048, 1024, 0)]:
    d_img = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    ts = []
    for _ in range(4):
        st = sc.render_panoramic_device(origin, W, H, stereo, d_img.data_ptr(), want_stats=True)
        ts.append(st["kernel_ms"])
    ms = min(ts)
    alg = st["nodes"] * 64 + st["tris"] * 76 + st["real_rays"] * 80
    line = "%dx%d stereo=%d, 10 spp, maxPathLength 16: kernel %.2f ms  %.0f Mrays/s  (rays %d, nodes/ray %.2f tris/ray %.2f, alg %.0f GB/s)" % (
        W, H, stereo, ms, st["real_rays"] / ms / 1e3, st["real_rays"], st["nodes"] / st["real_rays"], st["tris"] / st["real_rays"], alg / ms / 1e6)
    if W == 2048 and stereo == 1 and not os.environ.get("NO_CPU"):
        t0 = time.time()
        oimg, _, ost, _ = osc.render_panoramic(origin, W, H, stereo, 16, 10, O.RNG_HASH, seed=1)
        dt = time.time() - t0
        same = d_img.cpu().numpy().tobytes() == oimg.tobytes()
        line += "  | oracle on %d threads %.2f s = %.1f Mrays/s, image byte-equal: %s" % (len(os.sched_getaffinity(0)), dt, ost["real_rays"] / dt / 1e6, same)
    print(line, flush=True)
