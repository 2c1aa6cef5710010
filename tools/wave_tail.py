"""End-of-launch analysis of k_render_sm (diagnostic -DMGPU_UTIL build): when do the waves find the work cursor dry and
when do they end?  usage: python tools/wave_tail.py [world]   (rank 0's strips of a world-way split of the C2 frame)"""
import sys, os
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); sys.path.insert(0, _R); _os.chdir(_R)
os.environ["MGPU_WAVE_LOG"] = "1"
import numpy as np, torch
lib = os.path.join(_R, "mallie_amd", "libmallie_mgpu_util.so")
os.environ["MALLIE_MGPU_LIB"] = lib  # before the package is imported: the path is read at import
from mallie_amd import build as B
if not os.path.exists(lib) or os.environ.get("REBUILD"):
    B.build_variant(lib, ["-DMGPU_UTIL"])
import mallie_amd as M
from mallie_amd.frame import strip_rows
g = np.load("tests/golden/cornell_obj.npz")
sc = M.Scene(g["verts"].astype(np.float64), g["faces"], g["matIDs"], g["normals"], None)
W, H, mpl, spp = 1920, 1080, 5, 16
frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
plane = sc.plane()
for world in [int(a) for a in sys.argv[1:]] or [8]:
    n_rows = len(strip_rows(H, world, 0))
    buf = torch.empty((n_rows, W, 3), dtype=torch.float32, device="cuda")
    for i in range(4):
        st = sc.render_strips_device(frame, W, H, buf.data_ptr(), n_rows, y_first=0, strip_h=8, y_period=8 * world,
                                     maxPathLength=mpl, passes=spp, plane=plane, seed=1, want_stats=True)
    w = sc.debug_words()
    nw = int(w[27])
    wl = sc.wave_log(nw)
    xcc = (wl[:, 3] & 0xf).astype(np.int64)
    t0 = wl[:, 0].min()  # wall_clock64(): 100 MHz, device-wide base
    start, end = (wl[:, 0] - t0).astype(np.float64), (wl[:, 1] - t0).astype(np.float64)
    dry = np.where(wl[:, 2] > 0, (wl[:, 2] - t0).astype(np.float64), np.nan)
    rays = (wl[:, 3] >> 8).astype(np.float64)
    span = end.max()
    pc = lambda a: " ".join("%.3f" % x for x in np.nanpercentile(a / span, [0, 10, 50, 90, 100]))
    print("world %d: kernel %.3f ms, %d waves, span %.3f ms" % (world, st["kernel_ms"], nw, span / 1e5))
    print("  (fractions of the span; pctl 0/10/50/90/100)  start %s | dry %s | end %s" % (pc(start), pc(dry), pc(end)))
    print("  drain per wave (end - dry): %s ; waves that never saw dry: %d" % (pc(end - dry), int(np.isnan(dry).sum())))
    # machine utilisation: wave-time alive / (waves * span), and after the first dry
    first_dry = np.nanmin(dry)
    alive_after = np.clip(end - first_dry, 0, None).sum() / (nw * (span - first_dry))
    print("  first dry at %.3f of span; waves alive on average after that: %.3f; rays/wave min/med/max %d/%d/%d" % (
        first_dry / span, alive_after, rays.min(), np.median(rays), rays.max()))
    ar, act, plen = wl[:, 4].astype(np.float64), (wl[:, 5] & 0xffff).astype(np.float64), (wl[:, 5] >> 16).astype(np.float64)
    sn, stt, ss = (wl[:, 6] & 0xfffff).astype(np.float64), ((wl[:, 6] >> 20) & 0xfffff).astype(np.float64), (wl[:, 6] >> 40).astype(np.float64)
    q = lambda a: " ".join("%.0f" % x for x in np.percentile(a, [0, 10, 50, 90, 100]))
    print("  at dry: lanes alive %s ; mean pathLength of those %.2f ; rays traced after dry per wave %s (per alive lane %.2f)" % (
        q(act), plen.sum() / max(act.sum(), 1), q(ar), ar.sum() / max(act.sum(), 1)))
    print("  steps after dry: NODE %s | TRI %s | SHADE %s ; drain us per step (median wave) %.2f" % (
        q(sn), q(stt), q(ss), float(np.median((end - np.nan_to_num(dry)) / np.maximum(sn + stt + ss, 1))) / 100.0))
    long_ = (end - np.nan_to_num(dry)) > np.nanpercentile(end - dry, 90)
    print("  slowest 10%% of drains: lanes alive %.1f, rays after dry %.1f, steps N/T/S %.0f/%.0f/%.0f" % (
        act[long_].mean(), ar[long_].mean(), sn[long_].mean(), stt[long_].mean(), ss[long_].mean()))
    # per workgroup: the CU is free when its last wave ends
    wg = nw // 16 if nw >= 16 else 1
    e = end[: wg * 16].reshape(wg, 16)
    print("  workgroup end (max over its 16 waves): %s ; mean wave end within wg relative to wg end: %.3f" % (
        pc(e.max(1)), float((e / e.max(1, keepdims=True)).mean())))
