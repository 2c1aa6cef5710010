"""ctypes binding of the C ABI in include/mgpu.h (libmallie_mgpu.so)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("MALLIE_MGPU_LIB") or os.path.join(_HERE, "libmallie_mgpu.so")  # override: A/B builds

RNG_STREAM, RNG_TABLE, RNG_HASH = 0, 1, 2
PRECISION_FP64, PRECISION_FP32 = 0, 1

# byte-identical to the reference's BVHNode / Ray / Intersection (bvh_accel.h:10-30, common.h:78-83, intersection.h:6-24)
NODE_DT = np.dtype([("bmin", "<f8", 3), ("bmax", "<f8", 3), ("flag", "<i4"), ("axis", "<i4"), ("data", "<u4", 2)])
RAY_DT = np.dtype([("org", "<f8", 3), ("dir", "<f8", 3), ("invDir", "<f8", 3), ("dirSign", "<i4", 3), ("pad_", "<i4")])
ISECT_DT = np.dtype([("t", "<f8"), ("u", "<f8"), ("v", "<f8"), ("faceID", "<u4"), ("materialID", "<u4"), ("f0", "<u4"),
                     ("f1", "<u4"), ("f2", "<u4"), ("pad_", "<u4"), ("position", "<f8", 3),
                     ("geometricNormal", "<f8", 3), ("normal", "<f8", 3), ("tangent", "<f8", 3),
                     ("binormal", "<f8", 3), ("texcoord", "<f8", 2)])
assert NODE_DT.itemsize == 64 and RAY_DT.itemsize == 88 and ISECT_DT.itemsize == 184


class MgpuError(RuntimeError):
    def __init__(self, status, where, detail):
        super().__init__("%s failed with status %d: %s" % (where, status, detail))
        self.status = status


class Stats(C.Structure):
    _fields_ = [("trace_calls", C.c_uint64), ("real_rays", C.c_uint64), ("nodes", C.c_uint64), ("tris", C.c_uint64),
                ("paths", C.c_uint64), ("stack_overflow", C.c_uint64), ("kernel_ms", C.c_double),
                ("total_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class FrameStats(C.Structure):
    _fields_ = [("world", C.c_int), ("members", C.c_int), ("rccl_ranks", C.c_int), ("exchange_mode", C.c_int),
                ("frames", C.c_uint64), ("exchange_frames", C.c_uint64), ("exchange_ops_per_frame", C.c_uint64),
                ("exchange_ms", C.c_double), ("enqueue_calls", C.c_uint64), ("enqueue_ms", C.c_double)]


_lib = None


def lib_path():
    return _LIB_PATH


def load_library():
    """Loads libmallie_mgpu.so. Fails loudly when it has not been built: there is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch's ROCm wheels bundle their own libamdhip64/libhsa-runtime (same SONAME as
    # /opt/rocm's).  Importing torch FIRST makes this library bind to that copy; loading ours first would pull in the
    # system runtime as well and the second HSA instance then sees no GPU.  MALLIE_NO_TORCH=1 skips it (pure HIP use).
    if os.environ.get("MALLIE_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    # tests/emu builds the library's sources for the HOST against a wave64 emulator (a CPU / GPU differential for the tests).  It is
    # test infrastructure, never a fallback: the product refuses to load it unless the caller says it is a test.
    if "_emu" in os.path.basename(_LIB_PATH) and os.environ.get("MALLIE_ALLOW_EMULATOR", "0") != "1":
        raise ImportError("%s is the tests' wave emulator build, not the product library; mallie_amd has no CPU path "
                          "(tests set MALLIE_ALLOW_EMULATOR=1)" % _LIB_PATH)
    if not os.path.exists(_LIB_PATH):
        raise ImportError("%s is missing: build it with `python -m mallie_amd.build` (needs hipcc). mallie_amd has no "
                          "CPU fallback." % _LIB_PATH)
    L = C.CDLL(_LIB_PATH)
    vp, sz, u64, u32, i32, dbl = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int, C.c_double
    L.mgpu_abi_version.restype = i32
    L.mgpu_device_count.restype = i32
    L.mgpu_last_error.restype = C.c_char_p
    L.mgpu_status_string.restype = C.c_char_p
    L.mgpu_status_string.argtypes = [i32]
    L.mgpu_scene_create.argtypes = [vp, sz, vp, sz, vp, vp, vp, vp, sz, vp, vp, sz, i32, C.POINTER(vp)]
    L.mgpu_scene_create.restype = i32
    L.mgpu_scene_destroy.argtypes = [vp]
    L.mgpu_scene_set_precision.argtypes = [vp, i32]
    L.mgpu_scene_set_precision.restype = i32
    L.mgpu_scene_destroy.restype = i32
    L.mgpu_scene_bbox.argtypes = [vp, vp, vp]
    L.mgpu_scene_bbox.restype = i32
    L.mgpu_scene_device_bytes.argtypes = [vp]
    L.mgpu_scene_device_bytes.restype = sz
    L.mgpu_scene_device.argtypes = [vp]
    L.mgpu_scene_device.restype = i32
    L.mgpu_trace.argtypes = [vp, vp, sz, vp, vp, vp]
    L.mgpu_trace.restype = i32
    L.mgpu_trace_server_stats.argtypes = [vp, vp, vp, vp, vp]
    L.mgpu_trace_server_stats.restype = i32
    L.mgpu_trace_calls_measure.argtypes = [vp, vp, sz, i32, vp, vp, vp]
    L.mgpu_trace_calls_measure.restype = i32
    L.mgpu_trace_server_retire.argtypes = [vp]
    L.mgpu_trace_server_retire.restype = i32
    L.mgpu_render.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, u64, u32, vp, vp,
                              vp]
    L.mgpu_render.restype = i32
    L.mgpu_render_strips_device.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, u64,
                                            u32, vp, vp, vp, vp]
    L.mgpu_render_strips_device.restype = i32
    L.mgpu_render_frames_device.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, u64,
                                            u32, i32, vp, vp, vp, vp]
    L.mgpu_render_frames_device.restype = i32
    L.mgpu_hash_state.argtypes = [u64, u32, u32, vp]
    L.mgpu_stats_read.argtypes = [vp, vp, i32]
    L.mgpu_stats_read.restype = i32
    L.mgpu_debug_words.argtypes = [vp, vp]
    L.mgpu_debug_words.restype = i32
    L.mgpu_render_step.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, u64, u32, vp, vp, vp]
    L.mgpu_render_step.restype = i32
    L.mgpu_render_stream.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    L.mgpu_render_stream.restype = i32
    L.mgpu_scene_set_render_ahead.argtypes = [vp, i32]
    L.mgpu_scene_set_render_ahead.restype = i32
    L.mgpu_render_ahead_stats.argtypes = [vp, vp, vp]
    L.mgpu_render_ahead_stats.restype = i32
    L.mgpu_stream_stats.argtypes = [vp, vp, vp, vp, vp]
    L.mgpu_stream_stats.restype = i32
    L.mgpu_render_aov.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, u64, u32, vp, vp, vp]
    L.mgpu_render_aov.restype = i32
    L.mgpu_frame_create.argtypes = [vp, vp, i32, i32, i32, i32, i32, C.POINTER(vp)]
    L.mgpu_frame_create.restype = i32
    L.mgpu_frame_create_rank.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, i32, C.POINTER(vp)]
    L.mgpu_frame_create_rank.restype = i32
    L.mgpu_frame_unique_id.argtypes = [vp]
    L.mgpu_frame_unique_id.restype = i32
    L.mgpu_frame_destroy.argtypes = [vp]
    L.mgpu_frame_destroy.restype = i32
    L.mgpu_frame_render.argtypes = [vp, vp, i32, i32, vp, i32, u64, u32, C.POINTER(i32)]
    L.mgpu_frame_render.restype = i32
    L.mgpu_frame_render_batch.argtypes = [vp, vp, i32, i32, vp, i32, u64, u32, i32, vp]
    L.mgpu_frame_render_batch.restype = i32
    L.mgpu_frame_wait.argtypes = [vp, i32, vp, C.POINTER(vp)]
    L.mgpu_frame_wait.restype = i32
    L.mgpu_frame_set_readback.argtypes = [vp, i32]
    L.mgpu_frame_set_readback.restype = i32
    L.mgpu_frame_wait_host.argtypes = [vp, i32, C.POINTER(vp)]
    L.mgpu_frame_wait_host.restype = i32
    L.mgpu_frame_done_event_wait.argtypes = [vp, i32, vp]
    L.mgpu_frame_done_event_wait.restype = i32
    L.mgpu_frame_stats.argtypes = [vp, C.POINTER(FrameStats), i32]
    L.mgpu_frame_stats.restype = i32
    L.mgpu_frame_rows.argtypes = [i32, i32, i32, i32]
    L.mgpu_frame_rows.restype = i32
    L.mgpu_frame_plan.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, i32]
    L.mgpu_frame_plan.restype = i32
    L.mgpu_frame_block_plan.argtypes = [i32, i32, i32, i32, i32, i32, vp]
    L.mgpu_frame_block_plan.restype = i32
    L.mgpu_frame_last_error.restype = C.c_char_p
    L.mgpu_occupancy_read.argtypes = [vp, vp]
    L.mgpu_occupancy_read.restype = i32
    L.mgpu_render_panoramic.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, u64, u32, vp, vp,
                                        C.POINTER(Stats)]
    L.mgpu_render_panoramic.restype = i32
    L.mgpu_render_panoramic_device.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, u64, u32, vp, vp,
                                               vp, C.POINTER(Stats)]
    L.mgpu_render_panoramic_device.restype = i32
    L.mgpu_trace_device.argtypes = [vp, vp, sz, vp, vp, vp, C.POINTER(Stats)]
    L.mgpu_trace_device.restype = i32
    L.mgpu_debug_tile_order.argtypes = [vp, vp, vp, sz]
    L.mgpu_debug_tile_order.restype = i32
    L.mgpu_debug_wave_log.argtypes = [vp, vp, sz]
    L.mgpu_debug_wave_log.restype = i32
    L.mgpu_timing_enable.argtypes = [vp, i32]
    L.mgpu_timing_enable.restype = i32
    L.mgpu_timing_read.argtypes = [vp, C.POINTER(dbl), C.POINTER(i32)]
    L.mgpu_timing_read.restype = i32
    L.mgpu_tonemap_device.argtypes = [i32, vp, vp, sz, i32, vp, vp]
    L.mgpu_tonemap_device.restype = i32
    L.mgpu_probe_path.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, C.POINTER(i32)]
    L.mgpu_probe_path.restype = i32
    L.mgpu_camera_frame.argtypes = [vp, vp, vp, vp, dbl, i32, i32, vp]
    L.mgpu_camera_frame.restype = i32
    L.mgpu_bvh_build.argtypes = [vp, sz, vp, sz, dbl, i32, i32, i32, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), vp]
    L.mgpu_bvh_build.restype = i32
    L.mgpu_bvh_build_device.argtypes = [vp, sz, vp, sz, dbl, i32, i32, i32, i32, C.POINTER(vp), C.POINTER(sz),
                                        C.POINTER(vp), vp, C.POINTER(dbl)]
    L.mgpu_bvh_build_device.restype = i32
    L.mgpu_free.argtypes = [vp]
    L.mgpu_plane_from_bbox.argtypes = [vp, vp, vp]
    _lib = L
    return L


def _check(rc, where):
    if rc != 0:
        L = load_library()
        raise MgpuError(rc, where, (L.mgpu_last_error() or b"").decode() or L.mgpu_status_string(rc).decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def abi_version():
    return load_library().mgpu_abi_version()


def device_count():
    return load_library().mgpu_device_count()


def hash_state(seed, pass_, pixel):
    st = np.zeros(4, "<u4")
    load_library().mgpu_hash_state(seed, pass_, pixel, _p(st))
    return st


def camera_frame(eye, lookat, up=(0, 1, 0), quat=(0, 0, 0, 0), fov=45.0, width=512, height=512):
    """Camera::BuildCameraFrame (camera.cc:40-220): returns origin, corner, du, dv as 12 doubles (host code)."""
    f = np.zeros(12)
    _check(load_library().mgpu_camera_frame(_p(_c(eye, "<f8")), _p(_c(lookat, "<f8")), _p(_c(up, "<f8")),
                                            _p(_c(quat, "<f8")), float(fov), int(width), int(height), _p(f)),
           "mgpu_camera_frame")
    return f


def bvh_build(verts, faces, costTaabb=0.2, minLeaf=16, maxDepth=256, binSize=64, device=None):
    """BVHAccel::Build (bvh_accel.cc:445-482) with BVHBuildOptions defaults -> (nodes, indices, stats).
    device=None: the host builder; device=k: the device builder on GPU k (same bytes; stats gains 'device_ms')."""
    L = load_library()
    verts = _c(verts, "<f8").reshape(-1, 3)
    faces = _c(faces, "<u4").reshape(-1, 3)
    pn, pi, nn = C.c_void_p(), C.c_void_p(), C.c_size_t()
    st = (C.c_int * 3)()
    ms = C.c_double(0)
    if device is None:
        _check(L.mgpu_bvh_build(_p(verts), len(verts), _p(faces), len(faces), costTaabb, minLeaf, maxDepth, binSize,
                                C.byref(pn), C.byref(nn), C.byref(pi), st), "mgpu_bvh_build")
    else:
        _check(L.mgpu_bvh_build_device(_p(verts), len(verts), _p(faces), len(faces), costTaabb, minLeaf, maxDepth,
                                       binSize, int(device), C.byref(pn), C.byref(nn), C.byref(pi), st, C.byref(ms)),
               "mgpu_bvh_build_device")
    nodes = np.frombuffer(C.string_at(pn, 64 * nn.value), NODE_DT).copy()
    idx = np.frombuffer(C.string_at(pi, 4 * len(faces)), "<u4").copy()
    L.mgpu_free(pn)
    L.mgpu_free(pi)
    out = dict(maxTreeDepth=st[0], numLeafNodes=st[1], numBranchNodes=st[2])
    if device is not None:
        out["device_ms"] = ms.value
    return nodes, idx, out


TONEMAP_LINEAR_RGB8, TONEMAP_GAMMA22_BGRA8 = 0, 1


def tonemap_device(d_image_ptr, d_count_ptr, npix, mode, d_out_ptr, device=0, stream=None):
    """mgpu_tonemap_device: 1/count + the console (linear RGB8) or SDL (gamma 2.2 BGRA8) display transform, on device."""
    _check(load_library().mgpu_tonemap_device(device, d_image_ptr, d_count_ptr, npix, mode, d_out_ptr, stream),
           "mgpu_tonemap_device")


def plane_from_bbox(bmin, bmax):
    """Ground-plane coefficients as Render() derives them on its first call (render.cc:620-627)."""
    pl = np.zeros(4, "<f4")
    load_library().mgpu_plane_from_bbox(_p(_c(bmin, "<f8")), _p(_c(bmax, "<f8")), _p(pl))
    return pl


def frame_rows(H, strip_h, world, rank):
    """Rows of an H-row frame owned by `rank` of `world` with strips of strip_h rows (mgpu_frame_rows)."""
    return load_library().mgpu_frame_rows(H, strip_h, world, rank)


def frame_plan(W, H, strip_h, world, owner):
    """(local offsets, frame offsets, counts) in floats of rank `owner`'s strips, as the exchange walks them (mgpu_frame_plan)."""
    L = load_library()
    n = L.mgpu_frame_plan(W, H, strip_h, world, owner, None, None, None, 0)
    if n < 0:
        raise ValueError("bad frame geometry")
    lo, fo, cnt = np.zeros(n, "<u8"), np.zeros(n, "<u8"), np.zeros(n, "<u8")
    L.mgpu_frame_plan(W, H, strip_h, world, owner, _p(lo), _p(fo), _p(cnt), n)
    return lo, fo, cnt


def frame_block_plan(W, H, strip_h, world, owner, force_exchange=False):
    """mgpu_frame_block_plan as a dict: where rank `owner`'s strip buffer lands in rank 0's staging area and the copies that deal
    it to the frame (offsets / pitches in bytes except staging_off / msg_floats, which count floats)."""
    out = np.zeros(10, "<u8")
    if load_library().mgpu_frame_block_plan(W, H, strip_h, world, owner, 1 if force_exchange else 0, _p(out)) != 0:
        raise ValueError("bad frame geometry")
    keys = ("staging_off", "msg_floats", "dst_off", "dst_pitch", "src_pitch", "width", "height", "tail_dst_off", "tail_src_off", "tail_bytes")
    return {k: int(v) for k, v in zip(keys, out)}


def frame_unique_id():
    """128-byte RCCL communicator id for Frame.create_rank: make it on ONE rank and hand it to the others."""
    buf = np.zeros(128, "u1")
    rc = load_library().mgpu_frame_unique_id(_p(buf))
    if rc:
        raise MgpuError(rc, "mgpu_frame_unique_id", load_library().mgpu_frame_last_error().decode())
    return buf


class Frame:
    """Multi-GPU frames behind the C ABI (mgpu_frame_*): interleaved row strips, one RCCL exchange per frame, the float
    frame assembled in rank 0's HBM.  Frame(scenes, devices, ...) drives several GPUs from this process;
    Frame.create_rank(scene, device, rank, world, id, ...) is one rank of a process-per-GPU job."""

    def __init__(self, scenes=None, devices=None, W=0, H=0, strip_h=8, frames_in_flight=1, _handle=None, _keep=None):
        self.W, self.H = W, H
        self._keep = _keep if _keep is not None else list(scenes)
        if _handle is not None:
            self.h = _handle
            return
        L = load_library()
        hs = (C.c_void_p * len(scenes))(*[s.h for s in scenes])
        dv = (C.c_int * len(scenes))(*devices)
        h = C.c_void_p()
        rc = L.mgpu_frame_create(hs, dv, len(scenes), W, H, strip_h, frames_in_flight, C.byref(h))
        if rc:
            raise MgpuError(rc, "mgpu_frame_create", L.mgpu_frame_last_error().decode())
        self.h = h

    @classmethod
    def create_rank(cls, scene, device, rank, world, uid, W, H, strip_h=8, frames_in_flight=1):
        L = load_library()
        h = C.c_void_p()
        rc = L.mgpu_frame_create_rank(scene.h, device, rank, world, _p(_c(uid, "u1")) if uid is not None else None, W, H, strip_h,
                                      frames_in_flight, C.byref(h))
        if rc:
            raise MgpuError(rc, "mgpu_frame_create_rank", L.mgpu_frame_last_error().decode())
        return cls(W=W, H=H, _handle=h, _keep=[scene])

    def close(self):
        if getattr(self, "h", None):
            load_library().mgpu_frame_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cam, maxPathLength, passes, plane=None, seed=1, pass_base=0, rng_mode=RNG_HASH):
        """Enqueues one frame; returns the slot to wait for."""
        slot = C.c_int(-1)
        L = load_library()
        rc = L.mgpu_frame_render(self.h, _p(_c(cam, "<f8")), maxPathLength, passes, _p(_c(plane, "<f4")), rng_mode, seed, pass_base,
                                 C.byref(slot))
        if rc:
            raise MgpuError(rc, "mgpu_frame_render", L.mgpu_frame_last_error().decode())
        return slot.value

    def render_batch(self, cam, maxPathLength, passes, n_frames, plane=None, seed=1, pass_base=0, rng_mode=RNG_HASH):
        """mgpu_frame_render_batch: n_frames consecutive frames (frame i: passes pass_base + i * passes ...) rendered by one
        launch per GPU; returns their slots."""
        L = load_library()
        slots = (C.c_int32 * n_frames)()
        rc = L.mgpu_frame_render_batch(self.h, _p(_c(cam, "<f8")), maxPathLength, passes, _p(_c(plane, "<f4")), rng_mode, seed,
                                       pass_base, n_frames, slots)
        if rc != 0:
            raise MgpuError(rc, "mgpu_frame_render_batch", L.mgpu_frame_last_error().decode())
        return list(slots)

    def wait(self, slot, to_host=False):
        """Blocks until the slot's frame is complete. Returns the device pointer of rank 0's frame (0 on other ranks), or a
        host array (H x W x 3 float32) with to_host=True."""
        L = load_library()
        dev = C.c_void_p()
        host = np.zeros((self.H, self.W, 3), "<f4") if to_host else None
        rc = L.mgpu_frame_wait(self.h, slot, _p(host), C.byref(dev))
        if rc:
            raise MgpuError(rc, "mgpu_frame_wait", L.mgpu_frame_last_error().decode())
        return host if to_host else (dev.value or 0)

    def set_readback(self, on=True):
        """mgpu_frame_set_readback: every frame enqueued from now on is copied to a pinned host buffer of its slot behind its
        exchange, on a copy stream (under the next frame's kernel when frames_in_flight >= 2)."""
        rc = load_library().mgpu_frame_set_readback(self.h, 1 if on else 0)
        if rc:
            raise MgpuError(rc, "mgpu_frame_set_readback", load_library().mgpu_frame_last_error().decode())

    def wait_host(self, slot, copy=False):
        """mgpu_frame_wait_host: blocks until the slot's read-back is complete; returns the pinned host frame as an H x W x 3
        float32 array that ALIASES the slot's buffer (valid until the slot renders again), or a copy of it."""
        ptr = C.c_void_p()
        rc = load_library().mgpu_frame_wait_host(self.h, slot, C.byref(ptr))
        if rc:
            raise MgpuError(rc, "mgpu_frame_wait_host", load_library().mgpu_frame_last_error().decode())
        if not ptr.value:
            return None  # this process does not hold rank 0
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(self.H, self.W, 3))
        return a.copy() if copy else a

    def stats(self, reset=False):
        """mgpu_frame_stats: world, members of this process, rccl_ranks (read back from the communicator), exchange mode
        ("block" / "strips"), frames enqueued, and the timed exchange steps (frames, receives per frame, summed device ms)."""
        st = FrameStats()
        rc = load_library().mgpu_frame_stats(self.h, C.byref(st), 1 if reset else 0)
        if rc:
            raise MgpuError(rc, "mgpu_frame_stats", load_library().mgpu_frame_last_error().decode())
        d = {n: getattr(st, n) for n, _ in st._fields_}
        d["transport"] = "copy" if st.exchange_mode >= 2 else "rccl"
        d["exchange_mode"] = {0: "block", 1: "strips"}.get(st.exchange_mode % 2, str(st.exchange_mode))
        return d

    def stream_wait(self, slot, stream):
        rc = load_library().mgpu_frame_done_event_wait(self.h, slot, C.c_void_p(stream))
        if rc:
            raise MgpuError(rc, "mgpu_frame_done_event_wait", load_library().mgpu_frame_last_error().decode())


class Scene:
    """Device-resident scene: what Scene::Init leaves behind (Mesh arrays + BVH), uploaded and re-laid-out once."""

    def __init__(self, verts, faces, matIDs=None, normals=None, uvs=None, nodes=None, indices=None, mat_diffuse=None,
                 device=0):
        L = load_library()
        verts = _c(verts, "<f8").reshape(-1, 3)
        faces = _c(faces, "<u4").reshape(-1, 3)
        matIDs = _c(matIDs, "<u4")
        normals = _c(normals, "<f8") if normals is not None and np.size(normals) else None
        uvs = _c(uvs, "<f8") if uvs is not None and np.size(uvs) else None
        if nodes is None:
            nodes, indices, _ = bvh_build(verts, faces)
        nodes = _c(nodes, NODE_DT)
        indices = _c(indices, "<u4")
        mat_diffuse = _c(mat_diffuse, "<f8") if mat_diffuse is not None and np.size(mat_diffuse) else None
        nm = 0 if mat_diffuse is None else mat_diffuse.size // 3
        self.n_faces, self.n_nodes, self.device = len(faces), len(nodes), device
        h = C.c_void_p()
        _check(L.mgpu_scene_create(_p(verts), len(verts), _p(faces), len(faces), _p(matIDs), _p(normals), _p(uvs),
                                   _p(nodes), len(nodes), _p(indices), _p(mat_diffuse), nm, device, C.byref(h)),
               "mgpu_scene_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            load_library().mgpu_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def bbox(self):
        lo, hi = np.zeros(3), np.zeros(3)
        _check(load_library().mgpu_scene_bbox(self.h, _p(lo), _p(hi)), "mgpu_scene_bbox")
        return lo, hi

    def plane(self):
        return plane_from_bbox(*self.bbox())

    def device_bytes(self):
        return load_library().mgpu_scene_device_bytes(self.h)

    def device_index(self):
        return load_library().mgpu_scene_device(self.h)

    def trace(self, rays, want_stats=False):
        """Batched Scene::Trace. rays: (n,6) [org,dir] or a RAY_DT array. -> (ISECT_DT array, hit uint8 array[, stats])"""
        rays = np.asarray(rays)
        if rays.dtype != RAY_DT:
            r6 = _c(rays, "<f8").reshape(-1, 6)
            rays = np.zeros(len(r6), RAY_DT)
            rays["org"], rays["dir"] = r6[:, :3], r6[:, 3:]
        rays = np.ascontiguousarray(rays)
        out = np.zeros(len(rays), ISECT_DT)
        hit = np.zeros(len(rays), "u1")
        st = Stats()
        _check(load_library().mgpu_trace(self.h, _p(rays), len(rays), _p(out), _p(hit), C.byref(st)), "mgpu_trace")
        return (out, hit, st.as_dict()) if want_stats else (out, hit)

    def trace_calls(self, rays, per_call=1):
        """The same rays as `per_call`-ray mgpu_trace calls without statistics -- Scene::Trace as the reference calls it
        (per_call = 1: the resident server; 2..64: the submission queue).  -> (ISECT_DT array, hit uint8 array)"""
        rays = np.asarray(rays)
        if rays.dtype != RAY_DT:
            r6 = _c(rays, "<f8").reshape(-1, 6)
            rays = np.zeros(len(r6), RAY_DT)
            rays["org"], rays["dir"] = r6[:, :3], r6[:, 3:]
        rays = np.ascontiguousarray(rays)
        out = np.zeros(len(rays), ISECT_DT)
        hit = np.zeros(len(rays), "u1")
        L = load_library()
        base_r, base_o, base_h = rays.ctypes.data, out.ctypes.data, hit.ctypes.data
        for i in range(0, len(rays), per_call):
            n = min(per_call, len(rays) - i)
            _check(L.mgpu_trace(self.h, C.c_void_p(base_r + i * RAY_DT.itemsize), n, C.c_void_p(base_o + i * ISECT_DT.itemsize),
                                C.c_void_p(base_h + i), None), "mgpu_trace")
        return out, hit

    def trace_calls_measure(self, rays, threads=1):
        """mgpu_trace_calls_measure: the rays as one-ray mgpu_trace calls from `threads` native threads.
        -> (ISECT_DT array, hit uint8 array, calls per second)"""
        rays = np.asarray(rays)
        if rays.dtype != RAY_DT:
            r6 = _c(rays, "<f8").reshape(-1, 6)
            rays = np.zeros(len(r6), RAY_DT)
            rays["org"], rays["dir"] = r6[:, :3], r6[:, 3:]
        rays = np.ascontiguousarray(rays)
        out = np.zeros(len(rays), ISECT_DT)
        hit = np.zeros(len(rays), "u1")
        rate = C.c_double(0.0)
        _check(load_library().mgpu_trace_calls_measure(self.h, _p(rays), len(rays), int(threads), _p(out), _p(hit), C.byref(rate)),
               "mgpu_trace_calls_measure")
        return out, hit, rate.value

    def trace_server_stats(self):
        """-> dict(launches, calls, alive, device_us): the resident server of the one-ray callers (include/mgpu.h)"""
        a, b, c, d = C.c_uint64(0), C.c_uint64(0), C.c_int32(0), C.c_double(0.0)
        _check(load_library().mgpu_trace_server_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "mgpu_trace_server_stats")
        return dict(launches=a.value, calls=b.value, alive=bool(c.value), device_us=d.value)

    def trace_server_retire(self):
        _check(load_library().mgpu_trace_server_retire(self.h), "mgpu_trace_server_retire")

    def trace_device(self, d_rays_ptr, n, d_out_ptr, d_hit_ptr, stream=None, want_stats=False):
        """mgpu_trace_device: rays (88 B each), Intersection records (184 B) and hit bytes at raw device addresses."""
        st = Stats()
        _check(load_library().mgpu_trace_device(self.h, C.c_void_p(d_rays_ptr), n, C.c_void_p(d_out_ptr),
                                                C.c_void_p(d_hit_ptr), C.c_void_p(stream or 0),
                                                C.byref(st) if want_stats else None), "mgpu_trace_device")
        return st.as_dict() if want_stats else None

    def set_render_ahead(self, on=True):
        _check(load_library().mgpu_scene_set_render_ahead(self.h, 1 if on else 0), "mgpu_scene_set_render_ahead")

    def render_ahead_stats(self):
        h, m = C.c_uint64(), C.c_uint64()
        _check(load_library().mgpu_render_ahead_stats(self.h, C.byref(h), C.byref(m)), "mgpu_render_ahead_stats")
        return {"hits": h.value, "misses": m.value}

    def render(self, frame, W, H, maxPathLength=16, passes=1, plane=None, rng_mode=RNG_HASH, rng_states=None, seed=1,
               pass_base=0, window=None, image=None, count=None, want_stats=True):
        """mgpu_render into host buffers. -> (image HxWx3 float32, count HxW int32, stats dict; None with want_stats=False: the
        call then may be served by the render-ahead, set_render_ahead)"""
        frame = _c(frame, "<f8")
        x0, y0, x1, y1 = window if window is not None else (0, 0, W, H)
        if image is None:
            image = np.zeros((H, W, 3), "<f4")
        if count is None:
            count = np.zeros((H, W), "<i4")
        plane = _c(plane, "<f4")
        rng_states = _c(rng_states, "<u4")
        st = Stats()
        _check(load_library().mgpu_render(self.h, _p(frame[0:3]), _p(frame[3:6]), _p(frame[6:9]), _p(frame[9:12]), W, H,
                                          x0, y0, x1, y1, maxPathLength, passes, _p(plane), rng_mode, _p(rng_states),
                                          seed, pass_base, _p(image), _p(count), C.byref(st) if want_stats else None), "mgpu_render")
        return image, count, (st.as_dict() if want_stats else None)

    def render_stream(self, frame, W, H, maxPathLength=16, passes=1, plane=None, stream_state=None, count=None, want_states=False):
        """mgpu_render_stream: Render() in the reference's own serial random stream -> (image, count, stats, stream_state after,
        start states or None).  stream_state: 4 words (default: the reference's fresh seed)."""
        frame = _c(frame, "<f8")
        state = np.array((123456789, 362436069, 521288629, 88675123) if stream_state is None else stream_state, "<u4").copy()
        image = np.zeros((H, W, 3), "<f4")
        if count is None:
            count = np.zeros((H, W), "<i4")
        states = np.zeros((passes, H, W, 4), "<u4") if want_states else None
        st = Stats()
        _check(load_library().mgpu_render_stream(self.h, _p(frame[0:3]), _p(frame[3:6]), _p(frame[6:9]), _p(frame[9:12]), W, H,
                                                 maxPathLength, passes, _p(_c(plane, "<f4")), _p(state), _p(image), _p(count),
                                                 _p(states), C.byref(st)), "mgpu_render_stream")
        return image, count, st.as_dict(), state, states

    def stream_classes(self, W, H):
        out = np.zeros((H, W), "u1")
        L = load_library()
        L.mgpu_debug_stream_classes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _check(L.mgpu_debug_stream_classes(self.h, _p(out), W * H), "mgpu_debug_stream_classes")
        return out

    def stream_stats(self):
        """mgpu_stream_stats: the last render_stream call's resolution of the reference's random stream."""
        ms, cl, rt, un = C.c_double(), C.c_int(), C.c_uint64(), C.c_uint32()
        _check(load_library().mgpu_stream_stats(self.h, C.byref(ms), C.byref(cl), C.byref(rt), C.byref(un)), "mgpu_stream_stats")
        return {"resolve_ms": ms.value, "classified": bool(cl.value), "retries": rt.value, "uncertain_pixels": un.value}

    def render_aov(self, frame, W, H, kind, rng_mode=RNG_HASH, rng_states=None, seed=1, pass_base=0):
        """mgpu_render_aov: ShowNormal (kind 0) / ShowUV (kind 1) of the whole frame -> (image, stats)."""
        frame = _c(frame, "<f8")
        image = np.zeros((H, W, 3), "<f4")
        st = Stats()
        _check(load_library().mgpu_render_aov(self.h, _p(frame[0:3]), _p(frame[3:6]), _p(frame[6:9]), _p(frame[9:12]), W, H,
                                              kind, rng_mode, _p(_c(rng_states, "<u4")), seed, pass_base, _p(image), None,
                                              C.byref(st)), "mgpu_render_aov")
        return image, st.as_dict()

    def render_step(self, frame, W, H, step, maxPathLength=16, plane=None, rng_mode=RNG_HASH, rng_states=None, seed=1,
                    pass_base=0, count=None):
        """mgpu_render_step: ONE Render() call with its `step` argument -> (image, count, stats)."""
        frame = _c(frame, "<f8")
        image = np.zeros((H, W, 3), "<f4")
        if count is None:
            count = np.zeros((H, W), "<i4")
        st = Stats()
        _check(load_library().mgpu_render_step(self.h, _p(frame[0:3]), _p(frame[3:6]), _p(frame[6:9]), _p(frame[9:12]), W, H,
                                               step, maxPathLength, _p(_c(plane, "<f4")), rng_mode, _p(_c(rng_states, "<u4")),
                                               seed, pass_base, _p(image), _p(count), C.byref(st)), "mgpu_render_step")
        return image, count, st.as_dict()

    def render_panoramic(self, origin, W, H, stereo, maxPathLength=16, samples=10, rng_mode=RNG_HASH, rng_states=None,
                         seed=1, pass_base=0, window=None, image=None, count=None):
        """mgpu_render_panoramic into host buffers -> (image HxWx3 float32, count HxW int32, stats dict)"""
        origin = _c(origin, "<f8")
        x0, y0, x1, y1 = window if window is not None else (0, 0, W, H)
        image = np.zeros((H, W, 3), "<f4") if image is None else image
        count = np.zeros((H, W), "<i4") if count is None else count
        rng_states = _c(rng_states, "<u4")
        st = Stats()
        _check(load_library().mgpu_render_panoramic(self.h, _p(origin), W, H, x0, y0, x1, y1, maxPathLength, samples,
                                                    int(stereo), rng_mode, _p(rng_states), seed, pass_base, _p(image),
                                                    _p(count), C.byref(st)), "mgpu_render_panoramic")
        return image, count, st.as_dict()

    def render_panoramic_device(self, origin, W, H, stereo, d_image_ptr, maxPathLength=16, samples=10, rng_mode=RNG_HASH,
                                d_rng_states_ptr=None, seed=1, pass_base=0, window=None, d_count_ptr=None, stream=None,
                                want_stats=False):
        """mgpu_render_panoramic_device: window-local device image / count at raw addresses, asynchronous on `stream`."""
        origin = _c(origin, "<f8")
        x0, y0, x1, y1 = window if window is not None else (0, 0, W, H)
        st = Stats()
        _check(load_library().mgpu_render_panoramic_device(
            self.h, _p(origin), W, H, x0, y0, x1, y1, maxPathLength, samples, int(stereo), rng_mode,
            C.c_void_p(d_rng_states_ptr or 0), seed, pass_base, C.c_void_p(d_image_ptr), C.c_void_p(d_count_ptr or 0),
            C.c_void_p(stream or 0), C.byref(st) if want_stats else None), "mgpu_render_panoramic_device")
        return st.as_dict() if want_stats else None

    def stats_read(self, reset=True):
        """Running device work counters since the last reset (synchronises)."""
        st = Stats()
        _check(load_library().mgpu_stats_read(self.h, C.byref(st), 1 if reset else 0), "mgpu_stats_read")
        return st.as_dict()

    def occupancy(self):
        """Active-lane accounting of the render kernel since the last stats reset (mgpu_occupancy_read): dict with the raw
        counters and the three fractions node / tri / shade (None where nothing was booked)."""
        w = np.zeros(9, "<u8")
        _check(load_library().mgpu_occupancy_read(self.h, _p(w)), "mgpu_occupancy_read")
        k = ("node_trips", "node_lanes", "tri_trips", "tri_lanes", "shade_steps", "shade_lanes", "node_steps", "tri_steps")
        d = {n: int(v) for n, v in zip(k, w[:8])}
        d["sample_every"] = int(w[8] & 0xFFFFFFFF)
        frac = lambda lanes, trips: (lanes / (64.0 * trips)) if trips else None
        d["node_frac"], d["tri_frac"] = frac(d["node_lanes"], d["node_trips"]), frac(d["tri_lanes"], d["tri_trips"])
        d["shade_frac"] = frac(d["shade_lanes"], d["shade_steps"])
        return d

    def debug_words(self):
        w = np.zeros(32, "<u8")
        _check(load_library().mgpu_debug_words(self.h, _p(w)), "mgpu_debug_words")
        return w

    def tile_order(self, n_tiles):
        """(cost, order) of the last render launch's cost-ordered hand-out, n_tiles = ceil(w/8) * ceil(rows/8)."""
        cost, order = np.zeros(n_tiles, "<u4"), np.zeros(n_tiles, "<u4")
        _check(load_library().mgpu_debug_tile_order(self.h, _p(cost), _p(order), n_tiles), "mgpu_debug_tile_order")
        return cost, order

    def wave_log(self, n_waves):
        w = np.zeros((n_waves, 8), "<u8")
        _check(load_library().mgpu_debug_wave_log(self.h, _p(w), n_waves), "mgpu_debug_wave_log")
        return w

    def timing_enable(self, on=True):
        _check(load_library().mgpu_timing_enable(self.h, 1 if on else 0), "mgpu_timing_enable")

    def timing_read(self):
        """(sum of kernel ms, launches) since the last read; HIP events on the launch stream (synchronises)."""
        ms, n = C.c_double(0), C.c_int(0)
        _check(load_library().mgpu_timing_read(self.h, C.byref(ms), C.byref(n)), "mgpu_timing_read")
        return ms.value, n.value

    def probe_path(self, frame, W, H, px, py, start_state, maxPathLength=16, plane=None):
        """mgpu_probe_path: per-iteration records (n,16) of one eye path traced on the device."""
        rec = np.zeros((maxPathLength, 16))
        n = C.c_int(0)
        _check(load_library().mgpu_probe_path(self.h, _p(_c(frame, "<f8")), W, H, px, py, maxPathLength,
                                              _p(_c(plane, "<f4")), _p(_c(start_state, "<u4")), _p(rec), C.byref(n)),
               "mgpu_probe_path")
        return rec[: n.value]

    def set_precision(self, precision):
        """mgpu_scene_set_precision: "fp64" (default, bit-identical to the reference) or "fp32" (the fast mode)."""
        code = {"fp64": PRECISION_FP64, "fp32": PRECISION_FP32}.get(precision, precision)
        _check(load_library().mgpu_scene_set_precision(self.h, int(code)), "mgpu_scene_set_precision")

    def render_strips_device(self, frame, W, H, d_image_ptr, n_rows, x0=0, x1=None, y_first=0, strip_h=None,
                             y_period=None, maxPathLength=16, passes=1, plane=None, rng_mode=RNG_HASH,
                             d_rng_states_ptr=None, seed=1, pass_base=0, d_count_ptr=None, stream=None,
                             want_stats=False):
        """mgpu_render_strips_device: device pointers in, nothing crosses PCIe. Asynchronous unless want_stats."""
        frame = _c(frame, "<f8")
        x1 = W if x1 is None else x1
        strip_h = n_rows if strip_h is None else strip_h
        y_period = strip_h if y_period is None else y_period
        plane = _c(plane, "<f4")
        st = Stats() if want_stats else None
        _check(load_library().mgpu_render_strips_device(
            self.h, _p(frame), W, H, x0, x1, y_first, strip_h, y_period, n_rows, maxPathLength, passes, _p(plane),
            rng_mode, d_rng_states_ptr, seed, pass_base, d_image_ptr, d_count_ptr, stream,
            C.byref(st) if st is not None else None), "mgpu_render_strips_device")
        return st.as_dict() if st is not None else None

    def render_frames_device(self, frame, W, H, d_image_ptrs, n_rows, x0=0, x1=None, y_first=0, strip_h=None, y_period=None,
                             maxPathLength=16, passes=1, plane=None, rng_mode=RNG_HASH, d_rng_states_ptr=None, seed=1,
                             pass_base=0, d_count_ptrs=None, stream=None, want_stats=False):
        """mgpu_render_frames_device: len(d_image_ptrs) consecutive frames of `passes` passes each, as many of them per
        launch as the scratch budget holds; frame f renders passes pass_base + f * passes ... into d_image_ptrs[f]."""
        frame = _c(frame, "<f8")
        x1 = W if x1 is None else x1
        strip_h = n_rows if strip_h is None else strip_h
        y_period = strip_h if y_period is None else y_period
        plane = _c(plane, "<f4")
        n = len(d_image_ptrs)
        images = (C.c_void_p * n)(*d_image_ptrs)
        counts = (C.c_void_p * n)(*d_count_ptrs) if d_count_ptrs is not None else None
        st = Stats() if want_stats else None
        _check(load_library().mgpu_render_frames_device(
            self.h, _p(frame), W, H, x0, x1, y_first, strip_h, y_period, n_rows, maxPathLength, passes, _p(plane),
            rng_mode, d_rng_states_ptr, seed, pass_base, n, images, counts, stream,
            C.byref(st) if st is not None else None), "mgpu_render_frames_device")
        return st.as_dict() if st is not None else None
