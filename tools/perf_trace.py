"""k_trace (batched Scene::Trace) throughput: kernel time from HIP events for n rays (host buffers, so total_ms includes PCIe)."""
import sys, os, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, _os.path.join(_R, "tests")); _os.chdir(_R)
import numpy as np
import mallie_amd as M
import oracle_lib as O
from mallie_amd.scenes import suzanne_grid
n = int(os.environ.get("NRAYS", 4_000_000))
rng = np.random.default_rng(1)
c = O.load_golden("cornell_obj")
for name in sys.argv[1:] or ["c2"]:
    if name == "c2":
        verts, faces, mats, normals = c["verts"].astype(np.float64), c["faces"], c["matIDs"], c["normals"]
    elif name == "teapot":
        g = O.load_golden("teapot_obj")
        verts, faces, mats, normals = g["verts"].astype(np.float64), g["faces"], g["matIDs"], g["normals"]
    else:
        verts, faces, mats, normals = suzanne_grid(c["verts"], c["faces"], int(name[4:]))
    nodes, idx, _ = M.bvh_build(verts, faces, device=0 if len(faces) > 65536 else None)
    sc = M.Scene(verts, faces, mats, normals, None, nodes, idx)
    bmin, bmax = sc.bbox()
    ctr, ext = (np.array(bmin) + np.array(bmax)) / 2, (np.array(bmax) - np.array(bmin))
    for kind in ("incoherent", "camera"):
        if kind == "incoherent":
            org = ctr + (rng.random((n, 3)) - 0.5) * ext * 2.0
            d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        else:
            w = int(np.sqrt(n * 16 / 9)); h = n // w
            eye = ctr + np.array([0, 0, ext.max() * 1.5])
            px = (np.arange(w * h) % w) / w - 0.5; py = 0.5 - (np.arange(w * h) // w) / h
            tgt = ctr + np.stack([px * ext.max() * 1.2, py * ext.max() * 0.7, np.zeros(w * h)], 1)
            d = tgt - eye; d /= np.linalg.norm(d, axis=1, keepdims=True)
            org = np.broadcast_to(eye, d.shape)
        rays = np.concatenate([org, d], 1)
        sc.trace(rays[:1000])
        best = None
        for _ in range(3):
            out, hit, st = sc.trace(rays, want_stats=True)
            if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
        st = best
        nr = len(rays)
        print("%s %s: %d rays, kernel %.3f ms -> %.0f Mrays/s (hit %.0f%%, nodes/ray %.1f tris/ray %.1f), call total %.0f ms; bytes in+out %.0f GB/s" % (
            name, kind, nr, st["kernel_ms"], nr / st["kernel_ms"] / 1e3, 100.0 * hit.mean(), st["nodes"] / nr, st["tris"] / nr, st["total_ms"], nr * (88 + 184 + 1) / st["kernel_ms"] / 1e6), flush=True)
