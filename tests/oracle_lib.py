"""ctypes binding of oracle/libmallie_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU checker (see oracle/mallie_oracle.h).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module; the product package (mallie_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libmallie_oracle.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

NODE_DT = np.dtype([("bmin", "<f8", 3), ("bmax", "<f8", 3), ("flag", "<i4"), ("axis", "<i4"), ("data", "<u4", 2)])
HIT_DT = np.dtype([("hit", "<u4"), ("faceID", "<u4"), ("materialID", "<u4"), ("f0", "<u4"), ("f1", "<u4"),
                   ("f2", "<u4"), ("t", "<f8"), ("u", "<f8"), ("v", "<f8"), ("position", "<f8", 3),
                   ("geometricNormal", "<f8", 3), ("normal", "<f8", 3), ("texcoord", "<f8", 2)])
STATS_FIELDS = ("trace_calls", "real_rays", "nodes", "tris", "garbage_nodes", "garbage_hits", "paths", "max_stack")

RNG_STREAM, RNG_TABLE, RNG_HASH = 0, 1, 2
REFERENCE_SEED = (123456789, 362436069, 521288629, 88675123)  # render.cc:123-127, thread 0


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STATS_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in STATS_FIELDS}


def build():
    """(Re)build the oracle library with the committed recipe."""
    subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "mallie_oracle.c")
        path = os.environ.get("MALLIE_ORACLE_LIB", LIB_PATH)  # tests/test_sanitizers_cpu.py: the ASan + UBSan build of the same source
        if path == LIB_PATH and (not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src)):
            build()
        L = C.CDLL(path)
        vp, sz, u64, u32, i32, dbl = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int, C.c_double
        L.mo_bvh_build.argtypes = [vp, sz, vp, sz, dbl, i32, i32, i32, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), vp]
        L.mo_bvh_build.restype = i32
        L.mo_free.argtypes = [vp]
        L.mo_scene_create.argtypes = [vp, sz, vp, sz, vp, vp, vp, vp, sz, vp, vp, sz]
        L.mo_scene_create.restype = vp
        L.mo_scene_destroy.argtypes = [vp]
        L.mo_scene_bbox.argtypes = [vp, vp, vp]
        L.mo_plane_from_bbox.argtypes = [vp, vp, vp]
        L.mo_trace.argtypes = [vp, vp, sz, vp, vp]
        L.mo_trace.restype = i32
        L.mo_camera_frame.argtypes = [vp, vp, vp, vp, dbl, i32, i32, vp]
        L.mo_generate_ray.argtypes = [vp, dbl, dbl, vp]
        L.mo_xorshift128.argtypes = [vp]
        L.mo_xorshift128.restype = dbl
        L.mo_hash_state.argtypes = [u64, u32, u32, vp]
        L.mo_render.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, u64, u32, vp, vp, vp,
                                vp, i32]
        L.mo_render.restype = i32
        L.mo_render_step.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, u64, u32, vp, vp, vp, vp]
        L.mo_render_step.restype = i32
        L.mo_render_aov.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, u64, u32, vp, vp, vp]
        L.mo_render_aov.restype = i32
        L.mo_render_panoramic.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, u64, u32, vp, vp,
                                          vp, C.POINTER(Stats), i32]
        L.mo_render_panoramic.restype = i32
        L.mo_generate_env_ray.argtypes = [vp, i32, i32, i32, C.c_double, C.c_double, vp]
        L.mo_generate_env_ray.restype = None
        L.mo_tonemap.argtypes = [vp, vp, sz, i32, vp]
        L.mo_probe_path.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, C.POINTER(i32), vp]
        L.mo_probe_path.restype = i32
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def bvh_build(verts, faces, costTaabb=0.2, minLeaf=16, maxDepth=256, binSize=64):
    """BVHAccel::Build with BVHBuildOptions defaults (bvh_accel.h:33-43). Returns (nodes, indices, stats)."""
    verts = _c(verts, "<f8").reshape(-1, 3)
    faces = _c(faces, "<u4").reshape(-1, 3)
    pn, pi, nn = C.c_void_p(), C.c_void_p(), C.c_size_t()
    st = (C.c_int * 3)()
    rc = lib().mo_bvh_build(_p(verts), len(verts), _p(faces), len(faces), costTaabb, minLeaf, maxDepth, binSize,
                            C.byref(pn), C.byref(nn), C.byref(pi), st)
    if rc:
        raise RuntimeError("mo_bvh_build failed: %d" % rc)
    nodes = np.frombuffer(C.string_at(pn, 64 * nn.value), NODE_DT).copy()
    idx = np.frombuffer(C.string_at(pi, 4 * len(faces)), "<u4").copy()
    lib().mo_free(pn)
    lib().mo_free(pi)
    return nodes, idx, dict(maxTreeDepth=st[0], numLeafNodes=st[1], numBranchNodes=st[2])


class OracleScene:
    def __init__(self, verts, faces, matIDs=None, normals=None, uvs=None, nodes=None, indices=None, mat_diffuse=None):
        self.verts = _c(verts, "<f8").reshape(-1, 3)
        self.faces = _c(faces, "<u4").reshape(-1, 3)
        self.matIDs = _c(matIDs, "<u4")
        self.normals = _c(normals, "<f8") if normals is not None and np.size(normals) else None
        self.uvs = _c(uvs, "<f8") if uvs is not None and np.size(uvs) else None
        if nodes is None:
            nodes, indices, _ = bvh_build(self.verts, self.faces)
        self.nodes = _c(nodes, NODE_DT)
        self.indices = _c(indices, "<u4")
        self.mat_diffuse = _c(mat_diffuse, "<f8") if mat_diffuse is not None and np.size(mat_diffuse) else None
        nm = 0 if self.mat_diffuse is None else self.mat_diffuse.size // 3
        self.h = lib().mo_scene_create(_p(self.verts), len(self.verts), _p(self.faces), len(self.faces),
                                       _p(self.matIDs), _p(self.normals), _p(self.uvs), _p(self.nodes),
                                       len(self.nodes), _p(self.indices), _p(self.mat_diffuse), nm)
        if not self.h:
            raise RuntimeError("mo_scene_create failed")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().mo_scene_destroy(self.h)
                self.h = None
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def bbox(self):
        lo, hi = np.zeros(3), np.zeros(3)
        lib().mo_scene_bbox(self.h, _p(lo), _p(hi))
        return lo, hi

    def plane(self):
        """Plane coefficients as Render() derives them on its first call (render.cc:620-627)."""
        lo, hi = self.bbox()
        pl = np.zeros(4, "<f4")
        lib().mo_plane_from_bbox(_p(lo), _p(hi), _p(pl))
        return pl

    def trace(self, rays, stats=None):
        rays = _c(rays, "<f8").reshape(-1, 6)
        out = np.zeros(len(rays), HIT_DT)
        rc = lib().mo_trace(self.h, _p(rays), len(rays), _p(out), C.byref(stats) if stats is not None else None)
        if rc:
            raise RuntimeError("mo_trace failed: %d" % rc)
        return out

    def probe_path(self, frame, px, py, start_state, maxPathLength=16, plane=None):
        rec = np.zeros((maxPathLength, 16))
        n = C.c_int(0)
        rad = np.zeros(3)
        rc = lib().mo_probe_path(self.h, _p(_c(frame, "<f8")), px, py, maxPathLength, _p(_c(plane, "<f4")),
                                 _p(_c(start_state, "<u4")), _p(rec), C.byref(n), _p(rad))
        if rc:
            raise RuntimeError("mo_probe_path failed: %d" % rc)
        return rec[: n.value], rad

    def render(self, frame, W, H, maxPathLength=16, passes=1, plane=None, rng_mode=RNG_HASH, stream_state=None,
               rng_states=None, seed=1, pass_base=0, window=None, count=None, want_states=False, nthreads=0):
        frame = _c(frame, "<f8")
        x0, y0, x1, y1 = window if window is not None else (0, 0, W, H)
        image = np.zeros((H, W, 3), "<f4")
        if count is None:
            count = np.zeros((H, W), "<i4")
        plane = _c(plane, "<f4")
        rng_states = _c(rng_states, "<u4")
        states_out = np.zeros((passes, H, W, 4), "<u4") if want_states else None
        st = Stats()
        rc = lib().mo_render(self.h, _p(frame), W, H, x0, y0, x1, y1, maxPathLength, passes, _p(plane), rng_mode,
                             _p(stream_state), _p(rng_states), seed, pass_base, _p(image), _p(count), _p(states_out),
                             C.byref(st), nthreads)
        if rc:
            raise RuntimeError("mo_render failed: %d" % rc)
        return image, count, st.as_dict(), states_out


    def render_aov(self, frame, W, H, mode, rng_mode=RNG_HASH, stream_state=None, rng_states=None, seed=1, pass_base=0,
                   want_states=False):
        """mo_render_aov: ShowNormal (mode 0) / ShowUV (mode 1) for every pixel -> (image, stats, start states or None)."""
        image = np.zeros((H, W, 3), "<f4")
        rng_states = _c(rng_states, "<u4")
        states_out = np.zeros((H, W, 4), "<u4") if want_states else None
        st = Stats()
        rc = lib().mo_render_aov(self.h, _p(_c(frame, "<f8")), W, H, mode, rng_mode, _p(stream_state), _p(rng_states), seed,
                                 pass_base, _p(image), _p(states_out), C.byref(st))
        if rc:
            raise RuntimeError("mo_render_aov failed: %d" % rc)
        return image, st.as_dict(), states_out

    def render_step(self, frame, W, H, step, maxPathLength=16, plane=None, rng_mode=RNG_HASH, stream_state=None,
                    rng_states=None, seed=1, pass_base=0, count=None, want_states=False):
        """mo_render_step: ONE Render() call with its `step` argument (render.cc:657-696) -> (image, count, stats, states)."""
        image = np.zeros((H, W, 3), "<f4")
        if count is None:
            count = np.zeros((H, W), "<i4")
        rng_states = _c(rng_states, "<u4")
        states_out = np.zeros((H, W, 4), "<u4") if want_states else None
        st = Stats()
        rc = lib().mo_render_step(self.h, _p(_c(frame, "<f8")), W, H, step, maxPathLength, _p(_c(plane, "<f4")), rng_mode,
                                  _p(stream_state), _p(rng_states), seed, pass_base, _p(image), _p(count), _p(states_out),
                                  C.byref(st))
        if rc:
            raise RuntimeError("mo_render_step failed: %d" % rc)
        return image, count, st.as_dict(), states_out

    def render_panoramic(self, origin, W, H, stereo, maxPathLength=16, samples=10, rng_mode=RNG_HASH, stream_state=None,
                         rng_states=None, seed=1, pass_base=0, window=None, count=None, want_states=False, nthreads=0):
        """mo_render_panoramic: one RenderPanoramic() call -> (image, count, stats, per-pixel start states or None)."""
        origin = _c(origin, "<f8")
        x0, y0, x1, y1 = window if window is not None else (0, 0, W, H)
        image = np.zeros((H, W, 3), "<f4")
        if count is None:
            count = np.zeros((H, W), "<i4")
        rng_states = _c(rng_states, "<u4")
        states_out = np.zeros((H, W, 4), "<u4") if want_states else None
        st = Stats()
        rc = lib().mo_render_panoramic(self.h, _p(origin), W, H, x0, y0, x1, y1, maxPathLength, samples, int(stereo),
                                       rng_mode, _p(stream_state), _p(rng_states), seed, pass_base, _p(image), _p(count),
                                       _p(states_out), C.byref(st), nthreads)
        if rc:
            raise RuntimeError("mo_render_panoramic failed: %d" % rc)
        return image, count, st.as_dict(), states_out


def generate_env_ray(origin, W, H, stereo, u, v):
    r = np.zeros(6)
    lib().mo_generate_env_ray(_p(_c(origin, "<f8")), int(W), int(H), int(stereo), float(u), float(v), _p(r))
    return r


def tonemap(image, count, mode):
    image = _c(image, "<f4").reshape(-1, 3)
    count = _c(count, "<i4").reshape(-1)
    out = np.zeros((len(count), 3 if mode == 0 else 4), "u1")
    lib().mo_tonemap(_p(image), _p(count), len(count), mode, _p(out))
    return out


def camera_frame(eye, lookat, up=(0, 1, 0), quat=(0, 0, 0, 0), fov=45.0, width=512, height=512):
    f = np.zeros(12)
    lib().mo_camera_frame(_p(_c(eye, "<f8")), _p(_c(lookat, "<f8")), _p(_c(up, "<f8")), _p(_c(quat, "<f8")), float(fov),
                          int(width), int(height), _p(f))
    return f


def generate_ray(frame, u, v):
    r = np.zeros(6)
    lib().mo_generate_ray(_p(_c(frame, "<f8")), float(u), float(v), _p(r))
    return r


def hash_state(seed, pass_, pixel):
    st = np.zeros(4, "<u4")
    lib().mo_hash_state(seed, pass_, pixel, _p(st))
    return st


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_uvs(g):
    """facevarying_uvs of a mesh fixture: all-zero placeholder arrays (mesh_loader.cc:64-65) are stored as a flag."""
    if not g["has_uvs"]:
        return None
    return np.zeros((len(g["faces"]), 6)) if g["uvs_all_zero"] else g["uvs"]


def scene_from_golden(name, own_bvh=False, **kw):
    """OracleScene from a tests/golden mesh fixture. own_bvh=True rebuilds the tree with the oracle's builder."""
    g = load_golden(name)
    return OracleScene(g["verts"], g["faces"], g["matIDs"], g["normals"] if g["has_normals"] else None,
                       golden_uvs(g),
                       None if own_bvh else g["nodes"], None if own_bvh else g["indices"], **kw)
