"""Timings of the smaller entry points on one GPU: MGPU_RNG_STREAM (stream resolution + frame), ShowNormal / ShowUV."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); os.chdir(R)
import numpy as np
import mallie_amd as M
import oracle_lib as O
g = O.load_golden("cornell_obj")
sc = M.Scene(g["verts"].astype(np.float64), g["faces"], g["matIDs"], g["normals"], None, g["nodes"], g["indices"])
for W, H in ((512, 512), (1920, 1080)):
    frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    sc.render_stream(frame, W, H, 16, 1, sc.plane())
    t0 = time.perf_counter()
    img, cnt, st, state, _ = sc.render_stream(frame, W, H, 16, 1, sc.plane())
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    img2, _, st2 = sc.render(frame, W, H, 16, 1, sc.plane(), M.RNG_HASH, seed=1)
    dt2 = time.perf_counter() - t0
    print("%dx%d one Render() pass, maxPathLength 16: reference stream %.1f ms in all (frame kernel %.2f ms), hash seeding %.1f ms in all (kernel %.2f ms); hit pixels %d" % (
        W, H, 1e3 * dt, st["kernel_ms"], 1e3 * dt2, st2["kernel_ms"], int((img != 0).any(-1).sum())))
    for kind in (0, 1):
        sc.render_aov(frame, W, H, kind)
        a, s = sc.render_aov(frame, W, H, kind)
        print("  AOV %s: kernel %.3f ms (%d rays, %.0f Mrays/s)" % ("normal" if kind == 0 else "uv", s["kernel_ms"], s["real_rays"], s["real_rays"] / s["kernel_ms"] / 1e3))
