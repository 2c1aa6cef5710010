"""Soak test: many random views, GPU vs oracle, counting differing pixels (expected: none; the device's acos / sin / cos
differ from glibc's by <= 1 ulp, which can only matter through a flipped hit / miss decision)."""
import sys, os, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, _os.path.join(_R, "tests")); _os.chdir(_R)
import numpy as np
import mallie_amd as M
import oracle_lib as O
n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(2026)
tot_px = tot_diff = tot_rays = n_stream = n_retry = 0
for mesh in ("cornell_obj", "teapot_obj"):
    g = O.load_golden(mesh)
    sc = M.Scene(g["verts"].astype(np.float64), g["faces"], g["matIDs"], g["normals"] if g["has_normals"] else None, None, g["nodes"], g["indices"])
    osc = O.scene_from_golden(mesh)
    bmin, bmax = map(np.array, sc.bbox())
    ctr, ext = (bmin + bmax) / 2, (bmax - bmin).max()
    for v in range(n_views):
        W, H = int(rng.integers(40, 160)), int(rng.integers(30, 120))
        eye = ctr + rng.normal(size=3) * ext * rng.uniform(0.3, 2.5)
        la = ctr + rng.normal(size=3) * ext * 0.2
        fov = float(rng.uniform(20, 90))
        mpl, passes = int(rng.integers(1, 17)), int(rng.integers(1, 4))
        plane = osc.plane() if rng.random() < 0.7 else None
        frame = M.camera_frame(eye, la, fov=fov, width=W, height=H)
        img, _, st = sc.render(frame, W, H, mpl, passes, plane, M.RNG_HASH, seed=v, pass_base=v)
        oimg, _, ost, _ = osc.render(frame, W, H, mpl, passes, plane, O.RNG_HASH, seed=v, pass_base=v, nthreads=0)
        d = int((img != oimg).any(-1).sum())
        tot_px += W * H; tot_diff += d; tot_rays += st["real_rays"]
        assert st["real_rays"] == ost["real_rays"] or d, (mesh, v)
        if d:
            print("  view %d of %s: %d differing pixels (%dx%d, mpl %d)" % (v, mesh, d, W, H, mpl), flush=True)
        # the reference's own random stream, resolved across the chip (mgpu_stream.hip), every 3rd view: start states of every
        # (pass, pixel), the state left behind and the image against the oracle's run in that stream
        if v % 3 == 0:
            ostate = np.array(O.REFERENCE_SEED, "<u4")
            so, _, _, ostates = osc.render(frame, W, H, mpl, passes, plane, O.RNG_STREAM, stream_state=ostate, want_states=True)
            si, _, _, state, states = sc.render_stream(frame, W, H, mpl, passes, plane, want_states=True)
            ds = int((si != so).any(-1).sum()) + (0 if np.array_equal(states, ostates) and np.array_equal(state, ostate) else W * H)
            tot_px += W * H; tot_diff += ds; n_stream += 1; n_retry += sc.stream_stats()["retries"]
            if ds:
                print("  stream view %d of %s: %d differing pixels / states (%dx%d, mpl %d, %d passes)" % (v, mesh, ds, W, H, mpl, passes), flush=True)
        # panoramic from inside / around the scene every 4th view
        if v % 4 == 0:
            stereo = int(rng.integers(0, 2))
            origin = ctr + rng.normal(size=3) * ext * 0.4
            pi, _, _ = sc.render_panoramic(origin, W, H, stereo, 16, 10, M.RNG_HASH, seed=v)
            po, _, _, _ = osc.render_panoramic(origin, W, H, stereo, 16, 10, O.RNG_HASH, seed=v)
            dp = int((pi != po).any(-1).sum())
            tot_px += W * H; tot_diff += dp
            if dp:
                print("  pano view %d of %s: %d differing pixels" % (v, mesh, dp), flush=True)
print("soak: %d views per scene, %d pixels, %d real rays, %d frames in the reference's stream (retries of its resolution: %s): %d differing pixels" % (
    n_views, tot_px, tot_rays, n_stream, n_retry, tot_diff))
