import sys, os, time, json
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, _os.path.join(_R, "tests")); _os.chdir(_R)
import numpy as np, torch
import mallie_amd as M
g = np.load("tests/golden/cornell_obj.npz")
sc = M.Scene(g["verts"].astype(np.float64), g["faces"], g["matIDs"], g["normals"], None)
W, H, mpl, spp = 1920, 1080, 5, 16
if len(sys.argv) > 1: mpl = int(sys.argv[1])
frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
plane = sc.plane()
ts = []
for i in range(6):
    buf.fill_(float("nan"))  # a pixel the kernel skips must show in the checksum
    st = sc.render_strips_device(frame, W, H, buf.data_ptr(), H, maxPathLength=mpl, passes=spp, plane=plane, seed=1, want_stats=True)
    ts.append(st["kernel_ms"])
ms = float(np.median(ts[1:]))
print("%s blocks/cu=%s mpl=%d: kernel %.2f ms  %.0f Mrays/s  (rays %d nodes/ray %.2f tris/ray %.2f) checksum %.6f" % (
    os.environ.get("MALLIE_MGPU_LIB", "default"), os.environ.get("MGPU_RENDER_BLOCKS_PER_CU", "2"), mpl, ms, st["real_rays"] / ms / 1e3, st["real_rays"],
    st["nodes"] / st["real_rays"], st["tris"] / st["real_rays"], float(buf.double().sum().item())))
if os.environ.get("UTIL"):
    sc.render_strips_device(frame, W, H, buf.data_ptr(), H, maxPathLength=mpl, passes=spp, plane=plane, seed=1, want_stats=True)
    w = sc.debug_words()
    ns, nl, ts_, tl, outer, trl, shl, genl = [int(x) for x in w[8:16]]
    print("  wave-level: outer iters %d (ideal %d, trace-lane util %.3f), node steps %d (lane util %.3f), tri steps %d (lane util %.3f), new paths/iter %.1f" % (
        outer, st["real_rays"] // 64, trl / max(1, 64 * outer), ns, st["nodes"] / max(1, 64 * ns), ts_, st["tris"] / max(1, 64 * ts_), genl / max(1, outer)))
    print("  inner loops: NODE %d iterations (%.2f per step, %.1f of 64 lanes busy per iteration), TRI %d iterations (%.2f per step, %.1f lanes); SHADE steps %d with %.1f lanes" % (
        nl, nl / max(1, ns), st["nodes"] / max(1, nl), tl, tl / max(1, ts_), st["tris"] / max(1, tl), outer, shl / max(1, outer)))
    print("  per outer iter: node steps %.2f (ideal %.2f)  tri steps %.2f (ideal %.2f)" % (ns / outer, st["nodes"] / 64 / outer, ts_ / outer, st["tris"] / 64 / outer))
    if int(w[6]):
        hf, hs, hd, he = int(w[6]) & 0xffffffff, int(w[6]) >> 32, int(w[7]) & 0xffffffff, int(w[7]) >> 32
        print("  leaf hints: consulted for %d leaves (%.2f per ray) in %d TRI steps, dropped %d triangle tests (%.2f per ray, %.1f per consulted leaf), %d leaves dropped whole (%.1f%%)" % (
            hf, hf / st["real_rays"], hs, hd, hd / st["real_rays"], hd / max(1, hf), he, 100.0 * he / max(1, hf)))
    cn, ct, cs = [int(x) for x in w[16:19]]
    cb, nb = 0, 0
    hb = [int(x) for x in w[28:32]]
    if sum(hb):
        print("  SHADE steps by lanes waiting: <16: %.1f%%  16-35: %.1f%%  36-47: %.1f%%  >=48: %.1f%%" % tuple(100.0 * h / sum(hb) for h in hb))
    if cb:
        print("  bounce: %d steps, %.0f cyc/step, share of all body cycles %.1f%%" % (nb, cb / max(1, nb), 100.0 * cb / (cn + ct + cs + cb)))
    if cn:
        tot = cn + ct + cs + cb
        print("  cycle share: node %.1f%% (%.0f cyc/step)  tri %.1f%% (%.0f cyc/step)  shade %.1f%% (%.0f cyc/step)  [wave wall-clock cycles incl. interleaving]" % (
            100 * cn / tot, cn / max(1, ns), 100 * ct / tot, ct / max(1, ts_), 100 * cs / tot, cs / max(1, outer)))
    if trl or genl:
        print("  SHADE sub-bodies (wave level): miss tail ran in %d steps (%.0f cyc each), bounce in %d steps (%.0f cyc each), path start in %d steps; of %d SHADE steps" % (
            trl, int(w[20]) / max(1, trl), genl, int(w[22]) / max(1, genl), int(w[5]), outer))
    sub = [int(x) for x in w[19:25]]
    if sum(sub):
        names = ["1a normals+plane", "(unused)", "1b miss/bounce/accum", "2 hand-out", "3a new path", "3b arm"]
        print("  SHADE sub-parts (lane-0 samples): " + ", ".join("%s %.1f%%" % (n, 100.0 * v / sum(sub)) for n, v in zip(names, sub)))
    if int(w[27]):
        print("  per-wave loop cycles: avg %.1fM  max %.1fM  waves %d ; sum of body cycles per wave %.1fM" % (int(w[25]) / int(w[27]) / 1e6, int(w[26]) / 1e6, int(w[27]), (cn + ct + cs) / int(w[27]) / 1e6))
    if os.environ.get("MGPU_WAVE_LOG"):
        nw = int(w[27])
        wl = sc.wave_log(nw).astype(np.float64)
        t0 = wl[:, 0].min()
        start, end, rays, xcc = (wl[:, 0] - t0) / 1e6, (wl[:, 1] - t0) / 1e6, wl[:, 2], wl[:, 3]
        dur = end - start
        print("  waves %d: start min/max %.2f/%.2f M, end pctl 10/50/90/100: %s M, dur pctl: %s" % (nw, start.min(), start.max(), np.percentile(end, [10, 50, 90, 100]).round(1), np.percentile(dur, [10, 50, 90, 100]).round(1)))
        print("  rays per wave pctl 0/10/50/90/100: %s ; corr(rays,dur)=%.3f" % (np.percentile(rays, [0, 10, 50, 90, 100]).round(0), np.corrcoef(rays, dur)[0, 1]))
        for x in range(8):
            m = xcc == x
            if m.any(): print("   xcc %d: waves %d end median %.1f max %.1f rays/wave %.0f" % (x, m.sum(), np.median(end[m]), end[m].max(), rays[m].mean()))
        # per block
        bl = np.arange(nw) // (nw // 256 if nw >= 256 else 1)
o = sc.occupancy()
print("  occupancy (sampled every %d): NODE %.3f  TRI %.3f  SHADE %.3f; booked steps node %d tri %d shade %d; trips/step node %.2f tri %.2f" % (
    o["sample_every"], o["node_frac"] or 0, o["tri_frac"] or 0, o["shade_frac"] or 0, o["node_steps"], o["tri_steps"], o["shade_steps"],
    o["node_trips"] / max(1, o["node_steps"]), o["tri_trips"] / max(1, o["tri_steps"])))
