#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: the digest of BASELINE config C2's full frame (cornellbox_suzanne, 1920x1080, 16 spp, maxPathLength
5, plane on, eye (0,0,20), MGPU_RNG_HASH seed 1) as rendered by the oracle (oracle/mallie_oracle.c, pinned to the
reference).  Writes tests/golden/c2_1080p_16spp_digest.npz: sha256 of the float32 frame, the oracle's work counters, 16
sample rows.  ~1 minute on 8 cores.  Run from the repo root: python oracle/make_c2_digest.py"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
osc = O.scene_from_golden("cornell_obj", own_bvh=True)  # the oracle's own builder == reference tree (test_oracle_golden)
W, H, mpl, spp = 1920, 1080, 5, 16
frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
t = time.time()
img, cnt, st, _ = osc.render(frame, W, H, mpl, spp, osc.plane(), O.RNG_HASH, seed=1, nthreads=os.cpu_count())
print("oracle frame in %.1f s" % (time.time() - t), st)
rows = np.arange(0, H, H // 16)[:16]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "c2_1080p_16spp_digest.npz"),
                    sha256=np.frombuffer(hashlib.sha256(img.tobytes()).digest(), "u1"), W=W, H=H, maxPathLength=mpl, passes=spp, seed=1,
                    real_rays=st["real_rays"], trace_calls=st["trace_calls"], nodes=st["nodes"], tris=st["tris"], paths=st["paths"],
                    row_ids=rows, rows=img[rows], sum=float(img.astype(np.float64).sum()), count_min=int(cnt.min()), count_max=int(cnt.max()))
