#!/usr/bin/env python3
"""bench.py -- Mrays/s + ms/frame of the render hot path on N MI355X GPUs (BASELINE.json metric).

Workload (configs[1]): cornellbox_suzanne, 1920x1080, 16 spp, 4 bounces (maxPathLength 5), plane on,
eye (0,0,20) -> (0,0,0), per-(pixel,pass) seeding, seed 1.  One "step" = one whole frame: 16 passes of Render()
semantics accumulated on the device in one persistent-kernel launch per GPU (+ one RCCL gather of the row strips to
rank 0 when N > 1; there three frames are kept in flight on three streams so that the gather and the drain of one
launch overlap the next frames, see --frames-in-flight and DESIGN.md 6).  Step k renders passes [16k, 16k+16) -- a
progressive renderer's next frame, not the same frame again.  The scene (mesh arrays from tests/golden, BVH built by this
library's host builder) is resident in HBM before the timed region.  At N = 1 a frame is SURVEY.md 8(d)'s frame: the passes
AND one read-back -- every frame is copied to pinned host memory on a copy stream behind its kernel (mgpu_frame_set_readback),
which with two frames in flight runs under the next frame's kernel, and the caller takes frame k - 1 while frame k renders;
`value` / `ms_per_step` time that.  The same frames staying in HBM (the headline of rounds 1-3) and with a synchronous
read-back are timed beside it (`frame_resident_in_hbm`, `frame_with_synchronous_readback`).

"rays" = BVH traversals actually performed ("real" rays: primary + bounce rays up to and including a path's first
miss); the reference's post-miss continuation rays are finished analytically and are NOT counted.

Roofline: what bounds the dominant kernel on this workload is fp64 VALU issue (the 92 KB BVH lives in LDS; HBM carries
1 % of its peak), so `roofline.bound` = "valu": executed VALU wave-instructions per second against the issue peak, from
the rocprofv3 PMC passes committed under profiles/ -- used only when they were collected on the very library this run
loads (sha256 stamp), null otherwise.  The active-lane fraction of the kernel's bodies is measured in this run by an
instrumented pass (libmallie_mgpu_occ.so, not timed).  The SURVEY 8(d) HBM-priced figure is kept as
`algorithmic_vs_hbm`.  `extra_configs` carries single-GPU lines for the HBM-resident scenes C3 and C4.

Launch:  python bench.py [--gpus N --steps K --warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL across processes needs dmabuf IPC on these hosts (already exported on the GPU boxes; kept in case the launcher's
# environment was rebuilt): must be set before the HIP runtime loads
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# algorithmic bytes per event: fp64 reference-layout node, pre-gathered fp64 triangle + face id,
# ray in (org+dir) + hit out (t,u,v,id)
B_NODE, B_TRI, B_RAY = 64, 76, 80
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
N_SIMD, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, max clock
# a wave64 fp64 VALU instruction occupies its SIMD for 4 cycles (16 fp64 lanes per SIMD and cycle = the 78.6 TFLOP/s fp64
# vector peak of the guide); SQ_ACTIVE_INST_VALU counts in those 4-cycle units
VALU_PEAK_GINST = N_SIMD * CLOCK_HZ / 4 / 1e9
# nominal VALU instructions per trip of each body of k_render_sm (ISA of the shipped library): only used
# to weight the three measured lane fractions into one number
BODY_WEIGHT = {"node": 50.0, "tri": 80.0, "shade": 870.0}


def so_sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


_SELF_PMC = {"entries": {}, "log": [], "t0": None, "enabled": True}
# wall-clock budget of the counter passes bench.py runs itself (all workloads together); the headline workload goes first
SELF_PMC_BUDGET_S = float(os.environ.get("MALLIE_BENCH_SELF_PMC_BUDGET", "360"))


def self_pmc(workload):
    """No committed PMC pass has seen the library loaded now: collect the passes here (tools/pmc_collect.py -- rocprofv3 in processes
    of their own, AFTER the timed region, counters never together with tracing), once per workload and run.  Returns (entry or
    None, reason).  MALLIE_BENCH_SELF_PMC=0 switches it off."""
    if os.environ.get("MALLIE_BENCH_SELF_PMC", "1") == "0" or not _SELF_PMC["enabled"]:
        return None, "self-collection switched off (MALLIE_BENCH_SELF_PMC=0, --no-extras or N > 1)"
    if workload in _SELF_PMC["entries"]:
        return _SELF_PMC["entries"][workload]
    try:
        res = _self_pmc_collect(workload)
    except Exception as e:  # noqa: BLE001 -- a measurement aid must never take the bench line down
        res = (None, "collecting the counter passes failed: %r" % (e,))
    _SELF_PMC["entries"][workload] = res
    return res


def _self_pmc_collect(workload):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_collect
    if pmc_collect.rocprof() is None:
        res = (None, "no rocprofv3 on this box")
    else:
        if _SELF_PMC["t0"] is None:
            _SELF_PMC["t0"] = time.monotonic()
        out_dir = os.path.join(ROOT, "gpurun_out", "self_pmc")
        head = workload == "c2"
        # the headline workload: every pass + a kernel trace; the HBM-resident extras: traffic first, the SQ pass while the budget lasts
        e = pmc_collect.collect(workload, out_dir, frames=3 if workload != "c5" else 1, passes=("sq", "fetch", "write", "sq2") if head else ("fetch", "write", "sq", "sq2"),
                                trace=head, trace_frames=10, timeout=180, deadline=_SELF_PMC["t0"] + SELF_PMC_BUDGET_S, log=_SELF_PMC["log"])
        if e:
            res = (e, "collected by THIS run after its timed region (tools/pmc_collect.py: rocprofv3 --pmc, one process per pass, over "
                      "tools/pmc_workload.py %s; raw CSVs under gpurun_out/self_pmc/): %s" % (workload, "; ".join(l for l in _SELF_PMC["log"] if l.startswith(workload))))
        else:
            res = (None, "rocprofv3 passes left nothing to read: %s" % "; ".join(_SELF_PMC["log"][-4:]))
    return res


def self_pmc_dump(lib_path):
    """What the self-collected passes gave, in profiles/pmc_current.json's format, under gpurun_out/self_pmc/ (it travels back from the GPU box)."""
    try:
        got = {w: e for w, (e, _) in _SELF_PMC["entries"].items() if e}
        if not got:
            return
        from mallie_amd import build as _b
        with open(os.path.join(ROOT, "gpurun_out", "self_pmc", "pmc_current.json"), "w") as f:
            json.dump({"so_sha256": so_sha256(lib_path), "source_sha256": _b.source_digest(), "tag": "self_pmc", "workloads": got,
                       "log": _SELF_PMC["log"]}, f, indent=1, sort_keys=True)
    except Exception:  # noqa: BLE001 -- a record for the builder, nothing the bench line depends on
        pass


def pmc_for(workload, lib_path):
    """Per-FRAME PMC numbers of the dominant kernel (summed over its launches of one frame) on `workload`: from profiles/pmc_current.json
    (profiles/collect_pmc.sh + profiles/summarize_pmc.py) when those passes ran on the library loaded now, else collected by this run
    itself (self_pmc).  Returns (dict or None, reason)."""
    why_not = "no profiles/pmc_current.json"
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_current.json")) as f:
            d = json.load(f)
    except (OSError, ValueError):
        d = None
    if d is not None:
        from mallie_amd import build as _b
        stamp = None
        if d.get("so_sha256") == so_sha256(lib_path):
            stamp = "library sha256 %s" % d["so_sha256"][:16]
        elif d.get("source_sha256") and d.get("source_sha256") == _b.source_digest() and not _b.is_stale():
            stamp = "source digest %s (same sources and flags, library rebuilt elsewhere)" % d["source_sha256"][:16]
        w = d.get("workloads", {}).get(workload) if stamp else None
        if w:
            return w, "profiles/pmc_current.json (rocprofv3 --pmc, separate passes), %s" % stamp
        why_not = ("profiles/pmc_current.json has no entry for %s" % workload) if stamp else \
            "profiles/pmc_current.json was collected on another build of the library (sha256 of library and sources differ)"
    e, why = self_pmc(workload)
    return e, (why if e else "%s; %s" % (why_not, why))


def pmc_stale(workload):
    """The committed PMC numbers of `workload` WHATEVER library they were collected on, with their tag -- for the bench line of a library
    no PMC pass could see (no rocprofv3 on the box): reported beside a null `frac`, never as it."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_current.json")) as f:
            d = json.load(f)
        return d.get("workloads", {}).get(workload), d.get("tag")
    except (OSError, ValueError):
        return None, None


def hbm_bytes(p):
    """HBM bytes per frame: 2 x FETCH_SIZE (gfx950 correction of the guide) + WRITE_SIZE, both reported in KiB.  The committed
    calibration passes (profiles/pmc_current.json "calibration": known byte counts in this library's access patterns) confirm
    the factor 2 for 16-byte reads and give ~1 for writes; they are reported by profiles/*_summary.txt, not applied here."""
    if not p or "FETCH_SIZE" not in p or "WRITE_SIZE" not in p:
        return None
    return int(2 * p["FETCH_SIZE"] * 1024 + p["WRITE_SIZE"] * 1024)


def occupancy_pass(workload):
    """The instrumented library on one frame of `workload` in a process of its own (one library per process)."""
    occ_lib = os.path.join(ROOT, "mallie_amd", "libmallie_mgpu_occ.so")
    if not os.path.exists(occ_lib):
        return None
    env = dict(os.environ, MALLIE_MGPU_LIB=occ_lib)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, "-m", "mallie_amd.occupancy", workload], cwd=ROOT, env=env, capture_output=True,
                           text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        o = json.loads(line)
    except (subprocess.SubprocessError, IndexError, ValueError, OSError):
        return None
    out = {k: (round(o[k], 4) if o[k] is not None else None) for k in ("node_frac", "tri_frac", "shade_frac")}
    wsum = usum = 0.0
    for body, trips in (("node", o["node_trips"]), ("tri", o["tri_trips"]), ("shade", o["shade_steps"])):
        if o[body + "_frac"] is not None:
            wsum += trips * BODY_WEIGHT[body]
            usum += trips * BODY_WEIGHT[body] * o[body + "_frac"]
    out["weighted"] = round(usum / wsum, 4) if wsum else None
    out["source"] = ("instrumented pass in this run (libmallie_mgpu_occ.so = same sources + -DMGPU_OCC=1, one frame, not timed; ~1 step "
                     "in %d booked: node %d, tri %d, shade %d steps); active lanes / 64 per trip of each body's loop, `weighted` by "
                     "trips x nominal instructions per trip (%s)" % (o["sample_every"], o["node_steps"], o["tri_steps"], o["shade_steps"],
                                                                      BODY_WEIGHT))
    return out


def cpu_info():
    model, sockets, cores_per_socket = "unknown", 1, None
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"Model name:\s*(.+)", txt)
        if m:
            model = m.group(1).strip()
        m = re.search(r"Socket\(s\):\s*(\d+)", txt)
        if m:
            sockets = int(m.group(1))
        m = re.search(r"Core\(s\) per socket:\s*(\d+)", txt)
        if m:
            cores_per_socket = int(m.group(1))
    except (OSError, subprocess.SubprocessError):
        pass
    physical = sockets * cores_per_socket if cores_per_socket else None
    return model, physical


def cpu_baseline(cfg, frame, plane, mpl, pass_base, gpu_frame=None):
    """The oracle (this repo's CPU restatement, pinned bit-exact to the reference) timed on the host cores on a bounded
    sample of the same workload: whole 1920x1080 frames, same path length and seeding."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O  # the checker -- used here only as the timed CPU baseline
    from mallie_amd import workloads
    verts, faces, mats, normals = workloads.mesh_arrays(cfg)
    nodes, idx, _ = O.bvh_build(verts, faces)
    osc = O.OracleScene(verts, faces, mats, normals, None, nodes, idx)
    threads = len(os.sched_getaffinity(0))
    model, physical = cpu_info()
    W, H, spp = cfg["width"], cfg["height"], cfg["spp"]
    osc.render(frame, W, H, mpl, 1, plane, O.RNG_HASH, seed=cfg["seed"], window=(0, H // 2, W, H // 2 + 64), nthreads=threads)  # warm-up band
    done, dt, rays, same = 0, 0.0, 0, None
    chunk = spp if cfg["name"] == "C2" else 1  # C2: whole frames (the first one is compared with the GPU's); larger configs pass by pass
    while dt < 10.0 and done < 4 * spp:
        t0 = time.perf_counter()
        img, _, st, _ = osc.render(frame, W, H, mpl, chunk, plane, O.RNG_HASH, seed=cfg["seed"], pass_base=pass_base + done,
                                   nthreads=threads)
        dt += time.perf_counter() - t0
        if done == 0 and gpu_frame is not None and chunk == spp:
            # the first sample frame IS the last benchmarked frame (same seed and passes): the checker's image against the GPU's
            same = bool(img.tobytes() == gpu_frame.tobytes())
        rays += st["real_rays"]
        done += chunk
    return dict(gpu_frame_byte_equal=same, value=round(rays / dt / 1e6, 3), unit="Mrays/s", cores=threads, kind="port",
                cpu_model=model, physical_cores=physical, threads=threads,
                sample="%dx%d frames of the workload, %d passes in total (%d spp each), maxPathLength %d, OpenMP %d threads "
                       "(all logical CPUs of the host: %s physical cores, %s), %.1f s wall, %.0f ms/pass"
                       % (W, H, done, spp, mpl, threads, physical, model, dt, 1e3 * dt / done))


def cpu_reference(cfg, frame, plane):
    """Mallie's OWN OpenMP CPU path on this host (north_star: "next to Mallie's own OpenMP CPU path timed on the same box's host
    cores"): oracle/_ref/ref_driver = the unmodified reference sources compiled where they lie under /root/reference
    (oracle/Makefile) -- the binary travels to the GPU box, the sources do not.  It loads an .obj written here from the committed
    mesh arrays and times calls of mallie::Render() on all host threads.  The reference's kMaxPathLength is compiled in (16) and
    it counts nothing: the rays of a pass are counted by the oracle with the same path length (hash seeding; the reference's
    per-thread random streams give statistically the same paths).  Returns None when the binary is not there."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not os.path.exists(exe) or cfg.get("mesh") is None:
        return None
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from mallie_amd import workloads
    verts, faces, mats, normals = workloads.mesh_arrays(cfg)
    W, H = cfg["width"], cfg["height"]
    threads = len(os.sched_getaffinity(0))
    model, physical = cpu_info()
    with tempfile.TemporaryDirectory() as tmp:
        obj = os.path.join(tmp, "scene.obj")
        with open(obj, "w") as f:
            for v in verts:
                f.write("v %r %r %r\n" % (float(v[0]), float(v[1]), float(v[2])))
            for a, b, c in faces:
                f.write("f %d %d %d\n" % (a + 1, b + 1, c + 1))
        env = {k: v for k, v in os.environ.items() if k != "OMP_NUM_THREADS"}
        passes = 12
        args = [exe, "bench", "obj", obj, str(W), str(H), "1" if cfg["plane"] else "0", str(passes)] + \
               [repr(float(x)) for x in cfg["eye"]] + [repr(float(x)) for x in cfg["lookat"]]
        try:
            r = subprocess.run(args, cwd=tmp, env=env, capture_output=True, text=True, timeout=300)
        except (OSError, subprocess.SubprocessError) as e:
            return {"error": repr(e)}
        m = re.search(r"ref_bench passes=(\d+) seconds=([0-9.]+) threads=(\d+)", r.stdout)
        if r.returncode != 0 or not m:
            return {"error": "ref_driver bench failed (rc %d): %s" % (r.returncode, (r.stdout + r.stderr)[-300:])}
        per_pass = sorted(float(x) for x in re.findall(r"ref_bench_pass \d+ ([0-9.]+)", r.stdout))
    n, sec, thr = int(m.group(1)), float(m.group(2)), int(m.group(3))
    # rays of one reference pass: the oracle at the reference's path length (16), same frame, same plane
    nodes, idx, _ = O.bvh_build(verts, faces)
    osc = O.OracleScene(verts, faces, mats, normals, None, nodes, idx)
    _, _, st, _ = osc.render(frame, W, H, 16, 1, plane, O.RNG_HASH, seed=cfg["seed"], nthreads=threads)
    return dict(value=round(st["real_rays"] * n / sec / 1e6, 3), unit="Mrays/s", cores=thr, kind="reference",
                cpu_model=model, physical_cores=physical, threads=thr, ms_per_pass=round(1e3 * sec / n, 2),
                ms_per_pass_median=round(1e3 * per_pass[len(per_pass) // 2], 2), mtrace_calls_per_s=round(st["trace_calls"] * n / sec / 1e6, 2),
                sample="%d calls of the unmodified reference's mallie::Render() (oracle/_ref/ref_driver bench; OpenMP %d threads, %s physical "
                       "cores, %s), %dx%d, plane %s, kMaxPathLength 16 as compiled into the reference (the GPU workload stops at %d): %.2f s; rays "
                       "per pass = %d real rays (%d Trace() calls) counted by the oracle at the same path length"
                       % (n, thr, physical, model, W, H, "on" if cfg["plane"] else "off", cfg["bounces"] + 1, sec, st["real_rays"], st["trace_calls"]))


def flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass
    sys.stdout.flush()


def time_frames(scene, render, steps, sync):
    """ms per frame (wall), mean kernel ms (HIP events on the launch stream), work counters of `steps` frames."""
    scene.stats_read(reset=True)
    scene.timing_enable(True)
    t0 = time.perf_counter()
    for k in range(steps):
        render(k)
    sync()
    wall = time.perf_counter() - t0
    kernel_ms, launches = scene.timing_read()
    scene.timing_enable(False)
    st = scene.stats_read(reset=True)
    return 1e3 * wall / steps, kernel_ms / max(launches, 1), st


def hbm_roofline(key, st, steps, kms, lib_path):
    """`roofline` of an HBM-resident configuration (C3 / C4 / C5): measured HBM bytes (PMC passes of THIS library) over the
    kernel time against the HBM peak, the SURVEY 8(d) algorithmic bytes beside it."""
    alg = (st["nodes"] * B_NODE + st["tris"] * B_TRI + st["real_rays"] * B_RAY) / steps
    p, why = pmc_for(key, lib_path)
    traffic = hbm_bytes(p)
    wait = (round(p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"], 3) if p and p.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in p else None)
    return {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "achieved": round(traffic / (kms * 1e-3) / 1e9, 1) if traffic else None,
            "frac": round(traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "traffic": traffic, "kernel": "k_render_sm", "kernel_avg_ms": round(kms, 3), "pmc_source": why,
            "algorithmic_vs_hbm": {"bytes_per_launch": int(alg), "GBps": round(alg / (kms * 1e-3) / 1e9, 1),
                                   "ratio_to_peak": round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "valu_issue_busy": (round(p["SQ_ACTIVE_INST_VALU"] * 4.0 / (kms * 1e-3 * CLOCK_HZ * N_SIMD), 3)
                                if p and "SQ_ACTIVE_INST_VALU" in p else None),
            "wave_cycles_waiting": wait,
            "note": "BVH resident in HBM (wide 128-byte node records + 80-byte triangles through L1/L2, the top of the tree in an LDS "
                    "treelet). `achieved` = measured HBM bytes (2*FETCH_SIZE + WRITE_SIZE per frame) / kernel time of a frame; the "
                    "algorithmic bytes of SURVEY 8(d) are mostly L1/L2 hits. Neither HBM nor VALU issue is saturated (DESIGN.md 4.1, 7)."}


def one_ray_calls(M, verts, faces, mats, normals, frame, W, H, device, ref_scene, n=6000):
    """calls/s of one-ray mgpu_trace calls (mgpu_trace_calls_measure: native threads, no binding overhead in the clock) on
    `n` primary rays spread over the frame; the records of every mode must equal the batched kernel's."""
    rng = np.random.default_rng(7)
    px = rng.integers(0, W, n).astype(np.float64) + 0.5
    py = rng.integers(0, H, n).astype(np.float64) + 0.5
    f = np.asarray(frame, np.float64).reshape(4, 3)  # origin, corner, du, dv
    d = f[1][None, :] + px[:, None] * f[2][None, :] + py[:, None] * f[3][None, :] - f[0][None, :]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.hstack([np.tile(f[0], (n, 1)), d])
    ref, ref_hit = ref_scene.trace(rays)
    modes = (("resident_server", {}), ("submission_queue", {"MGPU_TRACE_SERVER": "0"}),
             ("launch_per_call", {"MGPU_TRACE_SERVER": "0", "MGPU_TRACE_QUEUE": "0"}))
    res = {"unit": "calls/s", "rays": n, "threads": [1, 4, 16]}
    same = True
    for name, env in modes:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            sc = M.Scene(verts, faces, mats, normals, None, device=device)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        rates = []
        for nt in (1, 4, 16):
            o, h, rate = sc.trace_calls_measure(rays, threads=nt)
            rates.append(round(rate))
            same = same and o.tobytes() == ref.tobytes() and h.tobytes() == ref_hit.tobytes()
        res[name] = rates
        if name == "resident_server":
            st = sc.trace_server_stats()
            res["server_device_us_per_call"] = round(st["device_us"], 2)
            res["server_launches"] = int(st["launches"])
        sc.close()
    res["records_equal_batched_kernel"] = bool(same)
    res["note"] = ("primary rays of the workload's camera, one mgpu_trace call per ray; resident_server = k_trace_server polling a mailbox in "
                   "mapped host memory (no launch per call), submission_queue = concurrent callers share one launch, launch_per_call = "
                   "a launch, two copies and a synchronisation per ray behind a mutex")
    return res


def extra_config(key, lib_path, torch, steps=5):
    """One single-GPU line for an HBM-resident BASELINE configuration (C3 / C4 / C5): frames rendered into HBM, HIP-event kernel
    time (summed over the launches of a frame), algorithmic bytes, and the counter-based HBM fraction when the committed PMC
    passes match this library."""
    import mallie_amd as M
    from mallie_amd import workloads
    cfg = workloads.CONFIGS[key]
    sc = workloads.make_scene(cfg)
    W, H, mpl, spp = cfg["width"], cfg["height"], cfg["bounces"] + 1, cfg["spp"]
    frame = workloads.camera(cfg)
    plane = sc.plane()
    buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")

    def render(k):
        sc.render_strips_device(frame, W, H, buf.data_ptr(), H, maxPathLength=mpl, passes=spp, plane=plane, seed=cfg["seed"],
                                pass_base=k * spp)
    for k in range(2):
        render(k)
    torch.cuda.synchronize()
    ms, kms, st = time_frames(sc, render, steps, torch.cuda.synchronize)
    rays = st["real_rays"] / steps
    out = {"config": workloads.describe(cfg, workloads.n_tris(cfg)),
           "n_gpus": 1, "steps": steps, "ms_per_frame": round(ms, 3), "kernel_avg_ms": round(kms, 3),
           "value": round(rays / ms / 1e3, 1), "unit": "Mrays/s", "rays_per_frame": int(rays),
           "nodes_per_ray": round(st["nodes"] / st["real_rays"], 3), "tris_per_ray": round(st["tris"] / st["real_rays"], 3),
           "device_MB": round(sc.device_bytes() / 1e6, 1),
           "roofline": hbm_roofline(key, st, steps, kms, lib_path)}
    ref16 = buf.clone()  # the last timed frame over the default tree
    if key in ("c3", "c4"):
        # the fast mode (fp32) on this configuration too: its time, and how far its frame is from the fp64 frame
        # of the same passes (buf holds the last timed frame).  Never the headline; north_star's tolerance is 1e-4.
        try:
            ref64 = ref16
            sc.set_precision("fp32")
            for k in range(2):
                render(k)
            torch.cuda.synchronize()
            ms_f, kms_f, st_f = time_frames(sc, render, steps, torch.cuda.synchronize)
            l2 = ((buf.double() - ref64.double()) / spp).pow(2).sum(-1).sqrt()
            out["fast_mode_fp32"] = {"ms_per_frame": round(ms_f, 3), "kernel_avg_ms": round(kms_f, 3), "speedup_vs_fp64": round(ms / ms_f, 3),
                                     "rms_per_pixel_l2_to_fp64_frame": float("%.3g" % float(l2.pow(2).mean().sqrt().item())),
                                     "pixels_moved_over_1e-3": float("%.3g" % float((l2 > 1e-3).double().mean().item())),
                                     "inside_north_star_1e-4": bool(float(l2.pow(2).mean().sqrt().item()) <= 1e-4)}
        except Exception as e:  # an extra line must never take the headline down
            out["fast_mode_fp32"] = {"error": repr(e)}
    # The same frames over the tree the reference's own builder makes with BVHBuildOptions::minLeafPrimitives = 8 instead of its
    # default 16 (bvh_accel.h:33-43; a caller's choice, not this library's): the walk is instruction-bound in proportion to the
    # nodes and triangles a ray visits, and with the scene in HBM smaller leaves trade ~2 triangle tests for ~1 node.
    # The last timed frame over the other tree is compared byte for byte with the one over the default tree.
    try:
        sc.close()
        sc = workloads.make_scene(cfg, min_leaf=8)
        for k in range(2):
            render(k)
        torch.cuda.synchronize()
        ms8, kms8, st8 = time_frames(sc, render, steps, torch.cuda.synchronize)
        out["min_leaf_primitives_8"] = {"ms_per_frame": round(ms8, 3), "kernel_avg_ms": round(kms8, 3), "speedup_vs_default_tree": round(ms / ms8, 3),
                                        "nodes_per_ray": round(st8["nodes"] / st8["real_rays"], 3), "tris_per_ray": round(st8["tris"] / st8["real_rays"], 3),
                                        "device_MB": round(sc.device_bytes() / 1e6, 1),
                                        "frame_equals_default_tree_frame": bool(torch.equal(buf, ref16))}
    except Exception as e:  # an extra line must never take the headline down
        out["min_leaf_primitives_8"] = {"error": repr(e)}
    sc.close()
    del buf, ref16
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of this node. Under torch.distributed.run: one rank per GPU (WORLD_SIZE wins). Without a launcher and "
                         "N > 1: this one process drives N devices through the C ABI's multi-GPU frame (mgpu_frame_create)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=("c2", "c3", "c4", "c5"), default="c2",
                    help="BASELINE configuration rendered as the headline workload (default c2 = configs[1], the one the metric is quoted on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the occupancy pass, the read-back timing and extra_configs")
    ap.add_argument("--exchange", choices=("rccl", "torch"), default="rccl",
                    help="N > 1: rccl = the C ABI's multi-GPU frame (RCCL called directly), torch = torch.distributed gather")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="frames rendered by one persistent launch per GPU (mgpu_frame_render_batch); default 1 at N=1, 8 at N>1")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames enqueued concurrently (own stream and buffers each); default 1 on one GPU -- kernel time "
                         "then is what rocprofv3 shows -- and 3 on N > 1, where the RCCL gather and the end of a launch "
                         "overlap the next frames' rendering")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import mallie_amd as M
    from mallie_amd import workloads
    from mallie_amd.frame import FrameRenderer

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MALLIE_FORCE_DEVICE0"):  # debugging aid: several ranks on one GPU (RCCL normally refuses this)
        local_rank = 0
    if not torch.cuda.is_available() or M.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # Two ways to N GPUs.  Under a launcher (WORLD_SIZE > 1): one process per GPU, `world` ranks.  Without one and --gpus N > 1:
    # THIS process drives N devices (mgpu_frame_create: ncclCommInitAll, one RCCL rank per device) -- `single` below; torch then
    # only provides the per-device synchronisation.
    # (MALLIE_BENCH_SINGLE=1: that code path with N = 1 -- what the one-GPU test box can exercise of it)
    single = env_world == 1 and (args.gpus > 1 or bool(os.environ.get("MALLIE_BENCH_SINGLE")))
    share_devices = os.environ.get("MGPU_FRAME_TRANSPORT") == "copy"  # test aid: ranks may share a device (include/mgpu.h)
    if single and args.gpus > M.device_count() and not share_devices:
        raise SystemExit("--gpus %d: only %d HIP device(s) visible" % (args.gpus, M.device_count()))
    world = args.gpus if single else env_world
    n_gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # debugging aid: MALLIE_FORCE_GATHER=1 sends a single-GPU run through the torch.distributed N > 1 code path
    force_gather = world == 1 and bool(os.environ.get("MALLIE_FORCE_GATHER"))
    if env_world > 1 or force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=rank, world_size=env_world, device_id=dev)
    multi_proc = env_world > 1

    key = args.config
    cfg = workloads.CONFIGS[key]
    W, H = cfg["width"], cfg["height"]
    mpl, spp = cfg["bounces"] + 1, cfg["spp"]
    n_tris = workloads.n_tris(cfg)
    # scene(s): BVH by this library's builder (host below 65 536 triangles, device above), resident in HBM before the timed region
    devices = [d % M.device_count() for d in range(world)] if single else [local_rank]
    scenes = [workloads.make_scene(cfg, device=d) for d in devices]
    scene = scenes[0]
    torch.cuda.set_device(local_rank)
    frame = workloads.camera(cfg)
    plane = scene.plane() if cfg["plane"] else None
    # N > 1: a GPU renders 1/N of the frame, and the end of a persistent launch (waves running dry one by one, ~0.35 ms) does not
    # shrink with it -- so several frames share a launch (mgpu_frame_render_batch) and twice that many are in flight, the
    # exchange of one batch under the launch of the next.  N = 1: one frame per launch, one in flight (SURVEY 8(d)'s frame).
    fpl = args.frames_per_launch if args.frames_per_launch > 0 else (1 if world == 1 else 8)
    if multi_proc and args.exchange != "rccl":
        fpl = 1  # the torch.distributed formulation renders frame by frame
    fif = args.frames_in_flight if args.frames_in_flight > 0 else (min(16, 2 * fpl) if fpl > 1 else (1 if world == 1 else 3))
    fpl = min(fpl, fif)
    fr = None
    if not single or world == 1:
        fr = FrameRenderer(scene, frame, W, H, mpl, spp, plane, cfg["seed"], rank, world, dev, frames_in_flight=fif,
                           force_collective=force_gather)
    # N > 1: the exchange goes through the C ABI's multi-GPU frame (mgpu_frame_*: RCCL called directly, see include/mgpu.h for
    # the two exchange modes); torch.distributed only carries the 128-byte communicator id and the barriers.
    # --exchange torch keeps the torch.distributed gather of mallie_amd/frame.py.  All ranks take the same path.
    cframe, exchange = None, "none"
    if world == 1 and os.environ.get("MGPU_FRAME_FORCE_EXCHANGE"):  # debugging aid: the C ABI's exchange path on one GPU
        cframe = M.Frame.create_rank(scene, local_rank, 0, 1, None, W, H, strip_h=8, frames_in_flight=fif)
        exchange = "mgpu_frame_* forced on one GPU (RCCL send/recv to self)"
    # N = 1, the headline: SURVEY 8(d)'s frame = the passes + ONE read-back.  The frame object (C ABI) copies every frame into a
    # pinned host buffer of its slot on a copy stream, behind the frame; with two slots the copy of frame k runs under the
    # kernel of frame k + 1, and the caller takes frame k - 1 (mgpu_frame_wait_host) while frame k renders.
    want_readback = not force_gather and not os.environ.get("MALLIE_BENCH_NO_READBACK")
    readback = False
    if world == 1 and cframe is None and want_readback:
        fif = max(fif, 2)
        cframe = M.Frame.create_rank(scene, local_rank, 0, 1, None, W, H, strip_h=8, frames_in_flight=fif)
        exchange = "none (one GPU); every frame read back to pinned host memory under the next frame's kernel"
    if single:
        cframe = M.Frame(scenes, devices, W, H, strip_h=8, frames_in_flight=fif)
        exchange = "mgpu_frame_* (C ABI), one process driving %d devices (ncclCommInitAll)" % world
    elif multi_proc:
        exchange = "torch.distributed gather + re-interleave (mallie_amd/frame.py)"
        if args.exchange == "rccl":
            uid = torch.zeros(128, dtype=torch.uint8, device=dev)
            ok = torch.ones(1, dtype=torch.int32, device=dev)
            try:
                if rank == 0:
                    uid.copy_(torch.from_numpy(M.frame_unique_id()))
                dist.broadcast(uid, src=0)
                cframe = M.Frame.create_rank(scene, local_rank, rank, world, uid.cpu().numpy(), W, H, strip_h=8, frames_in_flight=fif)
            except Exception as e:  # noqa: BLE001 -- whatever it is, every rank must learn about it
                sys.stderr.write("rank %d: mgpu_frame_create_rank failed (%r), falling back to the torch gather\n" % (rank, e))
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                exchange = "mgpu_frame_* (C ABI), one process per GPU (ncclCommInitRank)"
            elif cframe is not None:
                cframe.close()
                cframe = None

    # a frame ends with ONE read-back (behind the RCCL exchange when N > 1).  With the frame object every frame is copied
    # from rank 0's HBM to pinned host memory on a copy stream while later frames render; the caller takes the frames enqueued BEFORE
    # the latest render call once that call is out (mgpu_frame_wait_host; other ranks only wait for their strips to have left).
    if cframe is not None and want_readback:
        cframe.set_readback(True)
        readback = True
    pending = []
    taken = {"frames": 0, "last": None}  # read-back mode: host frames the caller has taken, and the latest (aliases pinned memory)

    def take(slots):
        for sl in slots:
            got = cframe.wait_host(sl)
            if got is not None:
                taken["last"] = got
            taken["frames"] += 1

    def make_room(n):
        """Read-back mode: the frame object has `fif` slots; before n more frames are enqueued the OLDEST frames in flight are taken
        (they have had the longest to finish, and the frames behind them keep the GPUs busy meanwhile) until the n fit."""
        while readback and pending and len(pending) + n > fif:
            take([pending.pop(0)])

    def render_frame(k):
        if cframe is not None:
            make_room(1)
            pending.append(cframe.render(frame, mpl, spp, plane, seed=cfg["seed"], pass_base=k * spp))
        else:
            fr.render(pass_base=k * spp)

    def render_frames(k0, n):
        if cframe is not None and n > 1:
            make_room(n)
            pending.extend(cframe.render_batch(frame, mpl, spp, n, plane, seed=cfg["seed"], pass_base=k0 * spp))
        else:
            for k in range(k0, k0 + n):
                render_frame(k)

    def finish_frames():
        if cframe is not None:
            if readback:
                take(list(pending))
            else:
                for slot in sorted(set(pending[-fif:])):
                    cframe.wait(slot)
            del pending[:]

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)
        if multi_proc:
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize(dev)

    batched = cframe is not None and fpl > 1
    if cframe is not None:
        # every slot of the frame object (own stream, own scratch: the pass planes are allocated on a stream's first launch) is used
        # once before the W warm-up steps, so that no allocation lands in the timed region however small W and K are
        for k0 in range(0, fif, fpl if batched else 1):
            render_frames(k0, min(fpl, fif - k0) if batched else 1)
        finish_frames()
        sync_all()
    for k0 in range(0, args.warmup, fpl if batched else 1):
        render_frames(k0, min(fpl, args.warmup - k0) if batched else 1)
    finish_frames()
    sync_all()
    flush_c_stdio()  # the communicators exist by now: whatever RCCL had to say goes out before the measurement
    for sc in scenes:
        sc.stats_read(reset=True)
        sc.timing_enable(True)
    if cframe is not None:
        cframe.stats(reset=True)
    taken["frames"] = 0
    sync_all()
    t0 = time.perf_counter()
    for k0 in range(0, args.steps, fpl if batched else 1):  # frame k = passes [k*spp, (k+1)*spp): the next 16 samples per pixel
        render_frames(k0, min(fpl, args.steps - k0) if batched else 1)
    finish_frames()
    sync_all()
    elapsed = time.perf_counter() - t0
    host_frames_taken = taken["frames"]
    last_host_frame = None if taken["last"] is None else taken["last"].copy()  # the last timed frame as the caller received it
    per_rank_kernel, launch_counts, sts = [], [], []
    for sc in scenes:
        kernel_ms, launches = sc.timing_read()
        sc.timing_enable(False)
        per_rank_kernel.append(kernel_ms / max(launches, 1))
        launch_counts.append(launches)
        sts.append(sc.stats_read(reset=True))
    st = sts[0]
    fstats = cframe.stats() if cframe is not None else None
    last_pass_base = (args.steps - 1) * spp
    # N > 1 checks itself: the last timed frame once more through the N-rank frame object, and the same passes by rank 0's GPU
    # alone (the single-GPU path, which the test-suite pins to the oracle) -- per-(pixel, pass) seeding makes them the same bytes
    same_as_one_gpu = None
    if cframe is not None and world > 1:
        slot = cframe.render(frame, mpl, spp, plane, seed=cfg["seed"], pass_base=last_pass_base)
        got = cframe.wait(slot, to_host=(rank == 0))
        if rank == 0:
            ref = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
            scene.render_strips_device(frame, W, H, ref.data_ptr(), H, maxPathLength=mpl, passes=spp, plane=plane, seed=cfg["seed"],
                                       pass_base=last_pass_base)
            torch.cuda.synchronize(dev)
            same_as_one_gpu = bool(got.tobytes() == ref.cpu().numpy().tobytes())
            del ref
        for sc in scenes:
            sc.stats_read(reset=True)

    # max elapsed over ranks, sum of work over ranks
    work = [float(sum(s[k] for s in sts)) for k in ("real_rays", "nodes", "tris", "trace_calls", "paths")]
    if multi_proc:
        red = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        wk = torch.tensor(work, dtype=torch.float64, device=dev)
        kern = torch.zeros(world, dtype=torch.float64, device=dev)
        kern[rank] = per_rank_kernel[0]
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        dist.all_reduce(wk, op=dist.ReduceOp.SUM)
        dist.all_reduce(kern, op=dist.ReduceOp.SUM)
        elapsed = float(red.item())
        work = [float(x) for x in wk.tolist()]
        per_rank_kernel = [float(x) for x in kern.tolist()]
    rays, nodes, tris, trace_calls, paths = work

    if rank == 0:
        lib_path = M.lib_path()
        # counter passes of its own (when the committed ones are of another build): single GPU, full runs only -- never under
        # --no-extras, which is how profiles/collect_pmc.sh runs this script UNDER rocprofv3
        _SELF_PMC["enabled"] = world == 1 and not args.no_extras
        ms_per_step = 1e3 * elapsed / args.steps
        value = rays / elapsed / 1e6
        # the dominant kernel (k_render_sm) on rank 0: algorithmic bytes of one launch / its mean duration
        n_launch = max(launch_counts[0], 1)
        alg_bytes_launch = (st["nodes"] * B_NODE + st["tris"] * B_TRI + st["real_rays"] * B_RAY) / n_launch
        kernel_avg_ms = per_rank_kernel[0]
        ksec = kernel_avg_ms * 1e-3
        # the workloads' scenes carry no material table (every hit multiplies by the default 0.5 grey): grey, i.e. ONE float
        # per pixel and pass in the planes, unless the three-channel kernel is forced
        scene_is_grey = os.environ.get("MGPU_GREY", "1") != "0"
        if key != "c2":
            # HBM-resident configurations: kernel time of a FRAME (a frame whose planes exceed 1 GiB takes several launches)
            roof = hbm_roofline(key, st, args.steps, kernel_avg_ms * n_launch / args.steps, lib_path) if world == 1 else \
                {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": None, "frac": None, "traffic": None,
                 "kernel": "k_render_sm", "kernel_avg_ms": round(kernel_avg_ms, 3), "pmc_source": "PMC passes are single-GPU"}
        else:
            p, why = pmc_for("c2", lib_path) if world == 1 else (None, "PMC passes are single-GPU")
            traffic = hbm_bytes(p)
            alg_gbs = alg_bytes_launch / ksec / 1e9 if ksec > 0 else 0.0
            # `achieved` / `frac`: the kernel's executed VALU wave-instructions per launch (counter pass) over THIS run's mean launch
            # duration (HIP events on the launch stream, timed region).  The same count over the kernel's average in the
            # rocprofv3 --kernel-trace --stats pass of the SAME library (committed under profiles/, or collected by this run) is kept
            # beside it as `frac_traced_profile`: the two durations must agree (a traced run is a few % slower).
            ginst = p["SQ_INSTS_VALU"] / ksec / 1e9 if p and "SQ_INSTS_VALU" in p and ksec > 0 else None
            traced_ms = p.get("traced_kernel_avg_ms") if p else None
            ginst_traced = p["SQ_INSTS_VALU"] / (traced_ms * 1e-3) / 1e9 if ginst and traced_ms else None
            roof = {"bound": "valu", "unit": "Ginst/s", "peak": round(VALU_PEAK_GINST, 1),
                    "achieved": round(ginst, 1) if ginst else None, "frac": round(ginst / VALU_PEAK_GINST, 4) if ginst else None,
                    "frac_over": "this run's HIP-event kernel time (%.3f ms per launch, %d launches timed)" % (kernel_avg_ms, launch_counts[0]),
                    "frac_traced_profile": round(ginst_traced / VALU_PEAK_GINST, 4) if ginst_traced else None,
                    "traced_kernel_avg_ms": round(traced_ms, 3) if traced_ms else None,
                    "traffic": traffic, "kernel": "k_render_sm", "kernel_avg_ms": round(kernel_avg_ms, 3), "pmc_source": why,
                    "valu": None, "hbm": None,
                    "algorithmic_vs_hbm": {"bytes_per_launch": int(alg_bytes_launch), "GBps": round(alg_gbs, 1),
                                           "ratio_to_peak": round(alg_gbs / HBM_PEAK_GBS, 4),
                                           "note": "SURVEY 8(d): nodes*64 + tris*76 + rays*80 over the kernel time against 8 TB/s. Not a "
                                                   "roofline here: this scene's BVH is staged in LDS and those bytes never leave the CU"},
                    "note": "bound = fp64 VALU issue: `achieved` = executed VALU wave-instructions (SQ_INSTS_VALU per frame = per launch here) / mean kernel "
                            "time of THIS run (HIP events on the launch stream); `peak` = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 fp64 "
                            "instruction. `valu.issue_busy` = SQ_ACTIVE_INST_VALU x 4 cycles / SIMD cycles; `lane_occupancy` = active lanes "
                            "per executed instruction, measured in this run; their product is the useful share of the issue peak."}
            if p and "SQ_ACTIVE_INST_VALU" in p and ksec > 0:
                roof["valu"] = {"insts_per_frame": int(p["SQ_INSTS_VALU"]),
                                "wave_insts_per_ray": round(p["SQ_INSTS_VALU"] / max(st["real_rays"] / n_launch, 1), 2),
                                "issue_busy": round(p["SQ_ACTIVE_INST_VALU"] * 4.0 / (ksec * CLOCK_HZ * N_SIMD), 3),
                                "salu_insts_per_frame": int(p["SQ_INSTS_SALU"]) if "SQ_INSTS_SALU" in p else None}
            if p is None and world == 1 and ksec > 0:
                # no PMC pass exists for the library loaded now and none could be collected here: `achieved` / `frac` stay null.  What the last stamped library's
                # instruction count would give over THIS run's kernel time is reported under its own name (the count moves by a few %
                # with the kernel's code; the figure is an indication, not a measurement of this library)
                sp, stag = pmc_stale("c2")
                if sp and "SQ_INSTS_VALU" in sp:
                    roof["unstamped_estimate"] = {"pmc_of": "profiles/pmc_current.json, tag %s: ANOTHER build of the library" % stag,
                                                  "SQ_INSTS_VALU_of_that_build": sp["SQ_INSTS_VALU"],
                                                  "frac_if_the_count_were_unchanged": round(sp["SQ_INSTS_VALU"] / ksec / 1e9 / VALU_PEAK_GINST, 4)}
            if traffic and ksec > 0:
                roof["hbm"] = {"measured_GBps": round(traffic / ksec / 1e9, 1), "frac_of_peak": round(traffic / ksec / 1e9 / HBM_PEAK_GBS, 4),
                               "fetch_bytes": int(2 * p["FETCH_SIZE"] * 1024), "write_bytes": int(p["WRITE_SIZE"] * 1024),
                               "needed_write_bytes": int((4 if scene_is_grey else 12) * W * H * spp),
                               "note": "2*FETCH_SIZE + WRITE_SIZE per frame (one launch) of k_render_sm; needed_write = the per-pass radiance "
                                       "planes it produces (one float per pixel and pass when the scene's materials are grey -- three equal "
                                       "channels -- else three); WRITE_SIZE is uncalibrated on gfx950 (profiles/README.md)"}
        conf = {"workload": workloads.describe(cfg, n_tris) + "; 1 step = 1 frame = the next %d passes per pixel "
                            "(pass_base advances by %d per step)" % (spp, spp),
                "parallelism": "replicated scene, interleaved 8-row strips x%d, 1 RCCL exchange/frame: %s" % (world, exchange)
                               if world > 1 else "single GPU, persistent-threads kernel",
                "frames_in_flight": fif, "frames_per_launch": fpl if batched else 1,
                "readback": ("every frame copied to pinned host memory (%.1f MB) behind its kernel%s, on a copy stream, and "
                             "taken by the caller while later frames render: %d of %d timed frames taken inside the timed region"
                             % (12e-6 * W * H, " and exchange" if world > 1 else "", host_frames_taken, args.steps)) if readback else "none: frames stay in HBM",
                "rays_per_frame": int(rays / args.steps), "trace_calls_per_frame": int(trace_calls / args.steps),
                "nodes_per_ray": round(nodes / rays, 3), "tris_per_ray": round(tris / rays, 3),
                "work_counters": ("nodes / tris per ray = the reference's node pops and TriangleIsect calls: exactly the oracle's counters for primary "
                                  "rays and batched traces, within 0.2 % of them for bounce rays (the tests' bound, tests/test_gpu_parity.py assert_same_work: "
                                  "bounce directions carry the <= 1 ulp difference between the device's and glibc's acos / sin / cos)"
                                  + ("; with the scene in LDS the leaf hints drop tests the ray provably cannot pass and book them as made (how many: "
                                     "tools/perf_hint_classes.py on a diagnostic build, DESIGN.md 4.1)" if key == "c2" else "")),
                "mtrace_calls_per_s": round(trace_calls / elapsed / 1e6, 2),
                # what the multi-GPU machinery says about itself: communicator size read back from RCCL (ncclCommCount), the
                # exchange step's device time per frame (HIP events on rank 0's communicator stream) and mean kernel time per
                # launch on every rank (a launch carries frames_per_launch frames)
                "rccl_ranks": fstats["rccl_ranks"] if fstats else (env_world if multi_proc else 0),
                # which HIP device every rank of THIS process drives; ranks share a device only under MGPU_FRAME_TRANSPORT=copy (a test
                # aid: no RCCL, no scaling -- the line then is about the machinery, not about N GPUs)
                "devices": devices, "ranks_share_a_device": len(set(devices)) < len(devices),
                "kernel_ms_per_launch_by_rank": [round(x, 3) for x in per_rank_kernel]}
        if same_as_one_gpu is not None:
            conf["frame_equals_single_gpu_frame"] = same_as_one_gpu
        if fstats:
            conf["transport"] = fstats["transport"]
            conf["exchange_mode"] = fstats["exchange_mode"]
            conf["exchange_recvs_per_frame"] = fstats["exchange_ops_per_frame"]
            conf["exchange_ms_per_frame"] = (round(fstats["exchange_ms"] / fstats["exchange_frames"], 4)
                                             if fstats["exchange_frames"] else None)
            conf["exchange_frames_timed"] = fstats["exchange_frames"]
            # one host thread enqueues every member's launches, events and exchange step: what a render call costs on the host
            conf["enqueue_ms_per_call"] = round(fstats["enqueue_ms"] / fstats["enqueue_calls"], 4) if fstats.get("enqueue_calls") else None
        out = {
            "metric": "Mrays/sec + ms/frame at 1920x1080, cornellbox_suzanne, 1/2/4/8 GPU",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            # what `value` / `ms_per_step` time (since round 4; rounds 1-3 timed "frame_resident_in_hbm", reported below)
            "headline_definition": "r4+: SURVEY 8(d)'s frame = all passes of a frame + its read-back to pinned host memory, frames in flight >= 2" if readback else "frames left in HBM",
            "config": conf,
            "roofline": roof,
        }
        verts, faces, mats, normals = workloads.mesh_arrays(cfg) if key == "c2" else (None, None, None, None)
        if world == 1 and not args.no_extras and key == "c2":
            occ = occupancy_pass("c2")
            roof["lane_occupancy"] = occ
            if occ and occ.get("weighted") and roof.get("valu"):
                roof["useful_frac"] = round(roof["valu"]["issue_busy"] * occ["weighted"], 4)
            # SURVEY 8(d)'s frame = passes + ONE read-back: the same frames, each followed by its copy to pinned host memory
            host = torch.empty((H, W, 3), dtype=torch.float32).pin_memory()
            fr1 = FrameRenderer(scene, frame, W, H, mpl, spp, plane, cfg["seed"], 0, 1, dev)

            def render_rb(k):
                host.copy_(fr1.render(pass_base=k * spp), non_blocking=True)
                torch.cuda.synchronize(dev)  # the caller owns the frame before the next one starts (mallie::Render semantics)
            render_rb(0)
            ms_rb, _, _ = time_frames(scene, render_rb, args.steps, lambda: torch.cuda.synchronize(dev))
            out["frame_with_synchronous_readback"] = {"ms_per_frame": round(ms_rb, 3), "value": round(rays / args.steps / ms_rb / 1e3, 2),
                                          "unit": "Mrays/s", "note": "same frames, one in flight, each followed by its 24.9 MB device-to-host copy "
                                          "(pinned memory) and a synchronisation before the next frame starts (rounds 1-3: frame_with_readback)"}
            ms_res, kms_res, _ = time_frames(scene, lambda k: fr1.render(pass_base=k * spp), args.steps, lambda: torch.cuda.synchronize(dev))
            out["frame_resident_in_hbm"] = {"ms_per_frame": round(ms_res, 3), "kernel_avg_ms": round(kms_res, 3),
                                            "value": round(rays / args.steps / ms_res / 1e3, 2), "unit": "Mrays/s",
                                            "note": "same frames, one in flight, no read-back: the frame stays in HBM (the headline of rounds 1-3)"}
            # the cost-ordered tile hand-out predicts a frame from an earlier one: the same frames without it (the switch is
            # read when a scene is created)
            os.environ["MGPU_TILE_ORDER"] = "0"
            scene0 = M.Scene(verts, faces, mats, normals, None, device=local_rank)
            del os.environ["MGPU_TILE_ORDER"]
            fr0 = FrameRenderer(scene0, frame, W, H, mpl, spp, plane, cfg["seed"], 0, 1, dev)
            fr0.render(pass_base=0)
            ms_no, _, _ = time_frames(scene0, lambda k: fr0.render(pass_base=k * spp), args.steps, lambda: torch.cuda.synchronize(dev))
            out["tile_order_off"] = {"ms_per_frame": round(ms_no, 3), "note": "MGPU_TILE_ORDER=0 (image-order hand-out), same frames"}
            del fr0
            scene0.close()
            # Scene::Trace as the reference calls it -- ONE ray per call from every OpenMP thread (scene.cc:253-315, render.cc:403):
            # primary rays of this camera as one-ray mgpu_trace calls from 1 / 4 / 16 native host threads, through the resident
            # server (default), the submission queue and a launch per call (switches read when a scene is created)
            try:
                out["scene_trace_one_ray_calls"] = one_ray_calls(M, verts, faces, mats, normals, frame, W, H, local_rank, scene)
            except Exception as e:  # an extra line must never take the headline down
                out["scene_trace_one_ray_calls"] = {"error": repr(e)}
            # the C ABI's exchange step priced on this one GPU (MGPU_FRAME_FORCE_EXCHANGE: the frame's strips are sent to
            # ourselves through RCCL), both exchange modes, 1080p (135 strips) -- device time of the step per frame
            out["exchange_on_one_gpu"] = forced_exchange_timing(M, scene, frame, W, H, mpl, spp, plane, cfg["seed"], local_rank, torch, dev)
            # the reference's OWN random stream (render.cc:116-168: one xorshift128 state walked through the frame in scanline order)
            # resolved on the device: one 1080p pass, 16 segments (the reference's compiled-in kMaxPathLength) -- the only mode whose
            # image IS the reference's image.  Chip-wide resolution (mgpu_stream.hip) against the round-3 one-workgroup walk.
            try:
                out["reference_stream_1080p"] = reference_stream_timing(scene, frame, W, H, plane)
            except Exception as e:  # an extra line must never take the headline down
                out["reference_stream_1080p"] = {"error": repr(e)}
            # the fast mode (MGPU_PRECISION_FP32, SURVEY 7 step 6 "report both"): the same frames in float.  Never `value`:
            # its frames are close to the reference's, not equal to them -- the distance is measured here, on the last frame.
            try:
                frf = FrameRenderer(scene, frame, W, H, mpl, spp, plane, cfg["seed"], 0, 1, dev)
                ref64 = frf.render(pass_base=last_pass_base).clone()
                torch.cuda.synchronize(dev)
                scene.set_precision("fp32")
                frf.render(pass_base=0)
                ms_f, kms_f, st_f = time_frames(scene, lambda k: frf.render(pass_base=k * spp), args.steps, lambda: torch.cuda.synchronize(dev))
                d = (frf.render(pass_base=last_pass_base).double() - ref64.double()) / spp
                torch.cuda.synchronize(dev)
                l2 = d.pow(2).sum(-1).sqrt()
                out["fast_mode_fp32"] = {
                    "ms_per_frame": round(ms_f, 3), "kernel_avg_ms": round(kms_f, 3), "kernel": "k_render_f32",
                    "value": round(st_f["real_rays"] / args.steps / ms_f / 1e3, 2), "unit": "Mrays/s", "dtype": "f32",
                    "distance_to_fp64_frame": {"rms_per_pixel_l2": float("%.3g" % float(l2.pow(2).mean().sqrt().item())),
                                               "pixels_moved_over_1e-3": float("%.3g" % float((l2 > 1e-3).double().mean().item())),
                                               "note": "pixel means of the last timed frame (%d spp); north_star's tolerance is 1e-4" % spp},
                    "note": "mgpu_scene_set_precision(MGPU_PRECISION_FP32): same algorithm, visiting order and random stream in float on a "
                            "float copy of the scene; NOT bit-identical to the reference and not the headline"}
            except Exception as e:  # an extra line must never take the headline down
                out["fast_mode_fp32"] = {"error": repr(e)}
            finally:
                scene.set_precision("fp64")
        gpu_frame = None
        if world == 1 and not args.no_cpu_baseline:
            # the last timed frame (pass_base of the last step), re-rendered after the timed region so that the extras above
            # cannot have touched it
            fr.render(pass_base=last_pass_base)
            torch.cuda.synchronize(dev)
            gpu_frame = fr.frame_buffer.detach().cpu().numpy()
            if last_host_frame is not None:  # what the caller took from pinned memory inside the timed region, against a fresh render
                conf["host_frame_equals_rerendered_frame"] = bool(last_host_frame.tobytes() == gpu_frame.tobytes())
        if world == 1 and not args.no_extras:
            fr = None
            torch.cuda.empty_cache()
            extras = {}
            for k2 in ("c3", "c4", "c5"):
                if k2 == key:
                    continue
                try:
                    extras[k2] = extra_config(k2, lib_path, torch, steps=3 if k2 == "c5" else 5)
                except Exception as e:  # an extra line must never take the headline down
                    extras[k2] = {"error": repr(e)}
            out["extra_configs"] = extras
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, frame, plane, mpl, last_pass_base, gpu_frame)
            ref = cpu_reference(cfg, frame, plane)
            if ref is not None:
                out["cpu_reference"] = ref
    else:
        out = None
    if cframe is not None:
        cframe.close()
    if multi_proc or force_gather:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()
    # RCCL prints a version banner through C stdio, which sits in libc's buffer when stdout is a pipe and would otherwise
    # come out at process exit, AFTER the result: flush it first, on every rank, so that the JSON line is the last line
    flush_c_stdio()
    if out is not None:
        self_pmc_dump(M.lib_path())
        print(json.dumps(out), flush=True)


def reference_stream_timing(scene, frame, W, H, plane, mpl=16):
    """mgpu_render_stream, one pass: wall ms of the stream resolution (first call = classification + whatever attempts its
    verification asked for; later calls continue the stream with the camera's classes cached) and of the round-3 serial kernel."""
    import hashlib
    state, res, first = None, [], None
    for k in range(5):
        img, _, st, state, _ = scene.render_stream(frame, W, H, mpl, 1, plane, stream_state=state)
        ss = scene.stream_stats()
        if k == 0:
            first, digest = ss, hashlib.sha256(img.tobytes()).hexdigest()
        else:
            res.append(ss["resolve_ms"])
    os.environ["MGPU_STREAM_SERIAL"] = "1"
    try:
        t0 = time.perf_counter()
        img_s, _, st_s, _, _ = scene.render_stream(frame, W, H, mpl, 1, plane)
        serial_ms = 1e3 * (time.perf_counter() - t0)
    finally:
        del os.environ["MGPU_STREAM_SERIAL"]
    return {"resolve_ms_per_pass": round(float(np.median(res)), 3), "resolve_ms_first_call": round(first["resolve_ms"], 3),
            "resolve_ms_later_calls": [round(x, 3) for x in res], "uncertain_pixels_per_pass": ss["uncertain_pixels"],
            "resolutions_repeated_in_5_calls": int(ss["retries"]), "serial_kernel_call_ms": round(serial_ms, 1),
            "first_frame_equals_serial_kernels": bool(hashlib.sha256(img_s.tobytes()).hexdigest() == digest),
            "frame_kernel_ms": round(st["kernel_ms"], 3), "maxPathLength": mpl,
            "note": "start state of every pixel of a %dx%d pass in the reference's serial xorshift128 stream, from its seed alone: "
                    "classification (cached per camera), the chain over the uncertain pixels in rounds of 128 with every possible hit "
                    "count traced at once, and a verification trace of every pixel; wall ms from the first kernel to the verdict. "
                    "serial_kernel_call_ms = the whole call with MGPU_STREAM_SERIAL=1 (one workgroup walks the chain)" % (W, H)}


def forced_exchange_timing(M, scene, frame, W, H, mpl, spp, plane, seed, device, torch, dev, frames=6):
    """Device time of the C ABI's exchange step with the frame's strips sent to ourselves, for both exchange modes."""
    res = {}
    old = {k: os.environ.get(k) for k in ("MGPU_FRAME_FORCE_EXCHANGE", "MGPU_FRAME_EXCHANGE")}
    try:
        os.environ["MGPU_FRAME_FORCE_EXCHANGE"] = "1"
        for mode in ("block", "strips"):
            os.environ["MGPU_FRAME_EXCHANGE"] = mode
            try:
                cf = M.Frame.create_rank(scene, device, 0, 1, None, W, H, strip_h=8, frames_in_flight=1)
                cf.wait(cf.render(frame, mpl, spp, plane, seed=seed, pass_base=0))
                cf.stats(reset=True)
                t0 = time.perf_counter()
                for k in range(frames):
                    cf.wait(cf.render(frame, mpl, spp, plane, seed=seed, pass_base=k * spp))
                torch.cuda.synchronize(dev)
                wall = time.perf_counter() - t0
                fs = cf.stats()
                res[mode] = {"exchange_ms_per_frame": round(fs["exchange_ms"] / max(fs["exchange_frames"], 1), 4),
                             "recvs_per_frame": fs["exchange_ops_per_frame"], "rccl_ranks": fs["rccl_ranks"],
                             "ms_per_frame_wall": round(1e3 * wall / frames, 3)}
                cf.close()
            except Exception as e:  # noqa: BLE001
                res[mode] = {"error": repr(e)}
        res["note"] = ("MGPU_FRAME_FORCE_EXCHANGE=1 on one GPU: the whole frame (%dx%d float RGB, %d strips of 8 rows) sent to rank 0 = "
                       "ourselves through RCCL; device time of the exchange step (HIP events on the communicator stream); one frame in flight"
                       % (W, H, (H + 7) // 8))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return res


if __name__ == "__main__":
    main()
