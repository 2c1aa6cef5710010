"""Fast mode vs fp64: C2 frame time and distance between the frames; teapot / grid32 too."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
from mallie_amd import workloads
def run(key, spp_override=None):
    cfg = dict(workloads.CONFIGS[key])
    if spp_override: cfg["spp"] = spp_override
    W, H, mpl, spp = cfg["width"], cfg["height"], cfg["bounces"] + 1, cfg["spp"]
    verts, faces, mats, normals = workloads.mesh_arrays(cfg)
    sc = M.Scene(verts, faces, mats, normals, None)
    cam = workloads.camera(cfg)
    plane = sc.plane() if cfg["plane"] else None
    out = {}
    for prec in ("fp64", "fp32"):
        sc.set_precision(prec)
        buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        def go(pb=0, stats=False):
            return sc.render_strips_device(cam, W, H, buf.data_ptr(), H, maxPathLength=mpl, passes=spp, plane=plane, seed=cfg["seed"], pass_base=pb, want_stats=stats)
        for _ in range(3): go()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 10
        for k in range(K): go(k * spp)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / K * 1e3
        st = go(0, True)
        out[prec] = (ms, buf.cpu().numpy().astype(np.float64) / spp, st)
    a, b = out["fp64"][1], out["fp32"][1]
    l2 = np.sqrt(((a - b) ** 2).sum(-1))
    s64, s32 = out["fp64"][2], out["fp32"][2]
    print("%s: fp64 %.3f ms (kernel %.3f), fp32 %.3f ms (kernel %.3f) = %.2fx; per-pixel L2 of the %d-spp mean: rms %.3g, max %.3g, pixels moved > 1e-4: %.4f %%, > 1e-3: %.4f %%; mean image %.6f vs %.6f; rays %d vs %d, nodes/ray %.3f vs %.3f, tris/ray %.3f vs %.3f"
          % (key, out["fp64"][0], s64["kernel_ms"], out["fp32"][0], s32["kernel_ms"], out["fp64"][0] / out["fp32"][0], spp, np.sqrt((l2 ** 2).mean()), l2.max(),
             100.0 * (l2 > 1e-4).mean(), 100.0 * (l2 > 1e-3).mean(), a.mean(), b.mean(), s64["real_rays"], s32["real_rays"],
             s64["nodes"] / s64["real_rays"], s32["nodes"] / s32["real_rays"], s64["tris"] / s64["real_rays"], s32["tris"] / s32["real_rays"]), flush=True)
for key in sys.argv[1:] or ["c2", "c4", "c3"]:
    run(key, 16 if key == "c3" else None)
