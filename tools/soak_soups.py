"""Soak test of the LDS-resident walk (leaf hints, shared leaves) on random geometry: triangle soups of 20 .. 1 800 triangles --
lattice coordinates with duplicates and zero-area triangles, free coordinates, long slivers, tiny and huge scales -- built by the
reference's builder, rendered from random views (hash-seeded, several passes, with and without the plane) on the GPU and by the
oracle: differing pixels and work counters."""
import sys, os, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, _os.path.join(_R, "tests")); _os.chdir(_R)
import numpy as np
import mallie_amd as M
import oracle_lib as O
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(4242)
tot_px = tot_diff = tot_rays = n_count_diff = 0
for s in range(n_scenes):
    kind = s % 5
    nt = int(rng.integers(20, 1800))
    nv = max(3, int(nt * rng.uniform(0.3, 1.5)))
    if kind == 0:      # lattice soup with duplicates and zero-area triangles
        verts = rng.integers(-8, 9, (nv, 3)).astype(np.float64) * 0.5
    elif kind == 1:    # free coordinates
        verts = rng.normal(size=(nv, 3)) * 3.0
    elif kind == 2:    # slivers: one coordinate squeezed
        verts = rng.normal(size=(nv, 3)) * np.array([5.0, 5.0, 1e-3]) + rng.integers(-2, 3, (nv, 3)) * np.array([0, 0, 1.0])
    elif kind == 3:    # tiny scale far from the origin
        verts = rng.normal(size=(nv, 3)) * 1e-3 + np.array([100.0, -50.0, 25.0])
    else:              # huge scale
        verts = rng.normal(size=(nv, 3)) * 1e5
    faces = rng.integers(0, nv, (nt, 3)).astype(np.uint32)
    if kind == 0:
        nd = nt // 5
        faces[rng.integers(0, nt, nd)] = faces[rng.integers(0, nt, nd)]
        faces[:2, 2] = faces[:2, 1]
    if kind in (1, 2, 3, 4):  # small triangles: connect nearby vertices so that leaves are spatially coherent
        order = np.argsort(verts[:, 0])
        base = rng.integers(0, max(1, nv - 8), nt)
        faces = order[(base[:, None] + rng.integers(0, 8, (nt, 3))) % nv].astype(np.uint32)
    mats = rng.integers(0, 3, nt).astype("u4")
    nodes, idx, _ = M.bvh_build(verts, faces)
    sc = M.Scene(verts, faces, mats, None, None, nodes, idx)
    osc = O.OracleScene(verts, faces, mats, None, None, nodes, idx)
    bmin, bmax = map(np.array, sc.bbox())
    ctr, ext = (bmin + bmax) / 2, float((bmax - bmin).max())
    for v in range(3):
        W, H = int(rng.integers(40, 120)), int(rng.integers(30, 90))
        eye = ctr + rng.normal(size=3) * ext * rng.uniform(0.4, 2.0)
        la = ctr + rng.normal(size=3) * ext * 0.2
        mpl, passes = int(rng.integers(1, 9)), int(rng.integers(1, 4))
        plane = osc.plane() if rng.random() < 0.5 else None
        frame = M.camera_frame(eye, la, fov=float(rng.uniform(20, 90)), width=W, height=H)
        img, _, st = sc.render(frame, W, H, mpl, passes, plane, M.RNG_HASH, seed=s, pass_base=v)
        oimg, _, ost, _ = osc.render(frame, W, H, mpl, passes, plane, O.RNG_HASH, seed=s, pass_base=v, nthreads=0)
        d = int((img.view("u4") != oimg.view("u4")).any(-1).sum())
        tot_px += W * H; tot_diff += d; tot_rays += st["real_rays"]
        if (st["nodes"], st["tris"]) != (ost["nodes"], ost["tris"]): n_count_diff += 1
        if d or st["real_rays"] != ost["real_rays"]:
            print("  scene %d (kind %d, %d triangles) view %d: %d differing pixels, rays %d vs %d" % (s, kind, nt, v, d, st["real_rays"], ost["real_rays"]), flush=True)
    sc.close()
print("soup soak: %d scenes x 3 views, %d pixels, %d real rays: %d differing pixels; %d of %d frames with node / triangle counters off the oracle's (bounce rays: a box test within an ulp of its threshold)" % (
    n_scenes, tot_px, tot_rays, tot_diff, n_count_diff, 3 * n_scenes))
