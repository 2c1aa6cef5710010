"""ON THE GPU BOX: ms per mgpu_render_stream call (the reference's own random stream resolved on the device), chip-wide vs serial."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests")); os.chdir(_R)
import numpy as np
import mallie_amd as M
import oracle_lib as O
g = O.load_golden("cornell_obj")
sc = M.Scene(g["verts"], g["faces"], g["matIDs"], g["normals"], None)
plane = sc.plane()
for (W, H) in ((512, 512), (1920, 1080)):
    frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    res = {}
    for mode in ("chip", "serial") if not os.environ.get("NO_SERIAL") else ("chip",):
        if mode == "serial": os.environ["MGPU_STREAM_SERIAL"] = "1"
        else: os.environ.pop("MGPU_STREAM_SERIAL", None)
        ts = []
        state = None
        for k in range(4 if mode == "chip" else 2):
            t0 = time.perf_counter()
            img, _, st, state, _ = sc.render_stream(frame, W, H, 16, 1, plane, stream_state=state)
            ts.append(1e3 * (time.perf_counter() - t0))
            if mode == "chip": print("      ", sc.stream_stats())
        res[mode] = (ts, st["kernel_ms"], img)
        print("%dx%d %-6s: calls (ms) %s ; frame kernel %.2f ms" % (W, H, mode, np.round(ts, 1), st["kernel_ms"]))
    if "serial" in res:
        print("   (first chip call classifies; later ones reuse the classes)")
