"""The Mallie-compatible C++ facade (include/mallie/*.h + libmallie_mgpu.so): a driver written against the reference's
header names compiles and links; Scene::Init's readers and BVHAccel::Build/Dump/Load reproduce what the reference makes
of the same files (CPU); Render/RenderPasses/Scene::Trace run on the GPU and agree with the C-ABI path."""
import os
import struct
import subprocess

import numpy as np
import pytest

import mallie_amd as M
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("facade") / "facade_driver")
    libdir = os.path.dirname(M.lib_path())
    cmd = ["g++", "-O1", "-std=c++11", "-pthread", "-I", os.path.join(ROOT, "include", "mallie"),
           os.path.join(ROOT, "tests", "cpp", "facade_driver.cc"), "-L", libdir, "-lmallie_mgpu",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def read_mesh(prefix):
    with open(prefix + ".mesh", "rb") as f:
        nv, nf, hn, hu = struct.unpack("<QQBB", f.read(18))
        d = dict(verts=np.frombuffer(f.read(24 * nv), "<f8").reshape(nv, 3),
                 faces=np.frombuffer(f.read(12 * nf), "<u4").reshape(nf, 3), matIDs=np.frombuffer(f.read(4 * nf), "<u4"))
        d["normals"] = np.frombuffer(f.read(72 * nf), "<f8").reshape(nf, 9) if hn else np.zeros((0, 9))
        d["uvs"] = np.frombuffer(f.read(48 * nf), "<f8").reshape(nf, 6) if hu else np.zeros((0, 6))
    with open(prefix + ".bvh", "rb") as f:
        (nn,) = struct.unpack("<Q", f.read(8))
        d["nodes"] = np.frombuffer(f.read(64 * nn), M.NODE_DT)
        (ni,) = struct.unpack("<Q", f.read(8))
        d["indices"] = np.frombuffer(f.read(4 * ni), "<u4")
    return d


def check_against_golden(got, name):
    g = O.load_golden(name)
    assert np.array_equal(got["verts"], g["verts"].astype(np.float64))
    assert np.array_equal(got["faces"], g["faces"]) and np.array_equal(got["matIDs"], g["matIDs"])
    if g["has_normals"]:
        assert got["normals"].tobytes() == g["normals"].tobytes()
    else:
        assert got["normals"].size == 0
    if g["has_uvs"]:
        assert got["uvs"].tobytes() == O.golden_uvs(g).astype(np.float64).tobytes()
    assert got["nodes"].tobytes() == g["nodes"].tobytes() and np.array_equal(got["indices"], g["indices"])


@pytest.mark.parametrize("obj,golden", [("quirks.obj", "objload_quirks"), ("nomtl.obj", "objload_nomtl")])
def test_obj_reader_reproduces_reference_loader(driver, tmp_path, obj, golden):
    objs = os.path.join(ROOT, "tests", "golden", "objs")
    prefix = str(tmp_path / "m")
    r = subprocess.run([driver, "mesh", "obj", obj, "1.0", prefix], cwd=objs, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    check_against_golden(read_mesh(prefix), golden)


@pytest.mark.parametrize("vox,golden", [("tiny.vox", "objload_vox_default"), ("tiny_rgba.vox", "objload_vox_rgba")])
def test_vox_reader_reproduces_reference_loader(driver, tmp_path, vox, golden):
    """MagicaVoxel input (MeshLoader::LoadMagicaVoxel): cube mesh, material ids (colour index 0 -> -1), BVH and all 256
    palette materials -- the generated default palette included -- equal what the reference makes of the same file."""
    objs = os.path.join(ROOT, "tests", "golden", "objs")
    prefix = str(tmp_path / "m")
    r = subprocess.run([driver, "mesh", "vox", vox, "1.0", prefix], cwd=objs, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Mallie:info\tmsg:Success to load .vox file" in r.stdout
    check_against_golden(read_mesh(prefix), golden)
    g = O.load_golden(golden)
    mats = np.fromfile(prefix + ".mat", "<f8").reshape(256, 3)
    assert mats.tobytes() == g["materials"].tobytes()
    if "default" in golden:
        assert 0xFFFFFFFF in g["matIDs"] and tuple(g["materials"][0]) == (0.0, 0.0, 0.0)  # palette entry 0, voxel index 0


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference's scene files (build container only)")
@pytest.mark.parametrize("kind,fname,golden", [("obj", "cornellbox_suzanne.obj", "cornell_obj"),
                                              ("eson", "cornellbox_suzanne.eson", "cornell_eson"),
                                              ("obj", "teapot.obj", "teapot_obj")])
def test_scene_init_on_reference_scenes(driver, tmp_path, kind, fname, golden):
    prefix = str(tmp_path / "m")
    r = subprocess.run([driver, "mesh", kind, fname, "1.0", prefix], cwd=REF, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Mallie:info\tmsg:Success to load" in r.stdout and "BVH statistics" in r.stdout
    check_against_golden(read_mesh(prefix), golden)


def test_scene_init_failure_is_reported_like_the_reference(driver, tmp_path):
    r = subprocess.run([driver, "mesh", "obj", "no_such_file.obj", "1.0", str(tmp_path / "x")], capture_output=True,
                       text=True)
    assert r.returncode == 3 and "Mallie:err\tmsg:Failed to load .obj file" in r.stdout


@pytest.mark.parametrize("obj", ["bad_index.obj", "no_vertices.obj"])
def test_obj_with_out_of_range_face_indices_is_refused(driver, tmp_path, obj):
    """A face naming a v / vn / vt entry the file does not have: the reference's loader reads out of bounds there; this
    reader refuses the file the way Scene::Init reports any failed load."""
    objs = os.path.join(ROOT, "tests", "golden", "objs")
    r = subprocess.run([driver, "mesh", "obj", obj, "1.0", str(tmp_path / "x")], cwd=objs, capture_output=True, text=True)
    assert r.returncode == 3 and "Mallie:err\tmsg:Failed to load .obj file" in r.stdout
    assert "face index out of range" in r.stderr


def test_env_rays_and_plane_intersect_match_the_reference(driver, tmp_path):
    """Camera::GenerateEnvRay / GenerateStereoEnvRay (camera.cc:242-329) and Plane::intersect (prim-plane.cc:8-44) of the
    facade, host code: bit for bit what the reference's own objects return for the same probes (tests/golden/boundary.npz,
    produced by oracle/ref_driver.cc from the unmodified sources)."""
    g = O.load_golden("boundary")
    uvp, outp = str(tmp_path / "uv.bin"), str(tmp_path / "env.bin")
    g["uv"].astype("<f8").tofile(uvp)
    r = subprocess.run([driver, "envrays", str(int(g["W"])), str(int(g["H"]))] + [repr(float(x)) for x in g["eye"]] +
                       [repr(float(x)) for x in g["lookat"]] + [uvp, outp], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.fromfile(outp, "<f8").reshape(-1, 12).tobytes() == g["env"].tobytes()
    rp, op = str(tmp_path / "pl.rays"), str(tmp_path / "pl.out")
    g["plane_rays"].astype("<f8").tofile(rp)
    r = subprocess.run([driver, "plane"] + [repr(float(x)) for x in g["plane"]] + [rp, op], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(op, "<f8").reshape(-1, 22)
    assert got.tobytes() == g["plane_out"].tobytes()
    assert 0 < got[:, 0].sum() < len(got) and np.all(got[:, 20] == 7)  # hits and misses; faceID is left alone


@pytest.fixture(scope="module")
def console(tmp_path_factory):
    """tests/cpp/console_driver.cc: a caller shaped like main_console.cc:57-79,104-111, against the forwarding headers."""
    out = str(tmp_path_factory.mktemp("console") / "console_driver")
    libdir = os.path.dirname(M.lib_path())
    cmd = ["g++", "-O1", "-std=c++11", "-pthread", "-I", os.path.join(ROOT, "include", "mallie"),
           os.path.join(ROOT, "tests", "cpp", "console_driver.cc"), "-L", libdir, "-lmallie_mgpu",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_console_shaped_caller_compiles_and_reports_a_missing_scene(console, tmp_path):
    r = subprocess.run([console, "obj", "no_such_file.obj", "64", "48", "1", "frame", str(tmp_path / "o.ppm")],
                       capture_output=True, text=True)
    assert r.returncode == 3 and "Mallie:err\tmsg:Failed to load .obj file" in r.stdout


def _read_ppm(path):
    raw = open(path, "rb").read()
    head, w_h, mx, body = raw.split(b"\n", 3)
    w, h = [int(x) for x in w_h.split()]
    assert head == b"P6" and mx == b"255" and len(body) == 3 * w * h
    return np.frombuffer(body, "u1").reshape(h, w, 3)


@pytest.mark.gpu
def test_console_shaped_caller_on_gpu(console, tmp_path):
    """The console driver's two frames (main_console.cc:57-79: Render + HDRToLDR; :104-111: stereo RenderPanoramic) through
    the facade: the 8-bit frames equal the oracle's image put through the oracle's HDRToLDR."""
    obj = str(tmp_path / "cornell_like.obj")
    _write_cornell_obj(obj)
    g = O.load_golden("cornell_obj")
    osc = O.OracleScene(g["verts"].astype(np.float64), g["faces"], np.full(len(g["faces"]), 0xFFFFFFFF, "u4"), g["normals"], None)
    W, H = 96, 64
    out = str(tmp_path / "frame.ppm")
    r = subprocess.run([console, "obj", obj, str(W), str(H), "1", "frame", out], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and "[Mallie] Console mode" in r.stdout and "[Mallie] Output" in r.stdout, r.stdout + r.stderr
    frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    oimg, ocnt, _, _ = osc.render(frame, W, H, 16, 1, osc.plane(), O.RNG_HASH, seed=1)  # the facade's defaults: seed 1, 16 segments
    assert np.array_equal(_read_ppm(out), O.tonemap(oimg, ocnt, 0).reshape(H, W, 3))
    out = str(tmp_path / "pano.ppm")
    r = subprocess.run([console, "obj", obj, str(W), str(H), "0", "pano", out], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    origin = O.camera_frame((0.0, 1.0, 4.0), (0, 0, 0), width=W, height=H)[:3]
    oimg, ocnt, _, _ = osc.render_panoramic(origin, W, H, 1, 16, 10, O.RNG_HASH, seed=1)
    assert np.array_equal(_read_ppm(out), O.tonemap(oimg, ocnt, 0).reshape(H, W, 3))


@pytest.mark.gpu
def test_facade_render_with_a_step_on_gpu(driver, tmp_path):
    """mallie::Render(..., step = 4) through the facade (render.cc:684-696): block-filled image and count += 3 per call,
    equal to the oracle's Render(step) with the facade's seeding (pass counter advancing per call)."""
    obj = str(tmp_path / "cornell_like.obj")
    _write_cornell_obj(obj)
    g = O.load_golden("cornell_obj")
    osc = O.OracleScene(g["verts"].astype(np.float64), g["faces"], np.full(len(g["faces"]), 0xFFFFFFFF, "u4"), g["normals"], None)
    W, H, calls, mpl, seed, step = 96, 64, 2, 6, 5, 4
    out = str(tmp_path / "step.bin")
    r = subprocess.run([driver, "step", "obj", obj, str(W), str(H), "1", str(calls), str(mpl), str(seed), str(step), out],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(out, "<f4")
    img, count = raw[: 3 * W * H].reshape(H, W, 3), raw[3 * W * H:].view("<i4").reshape(H, W)
    frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    ocount = np.zeros((H, W), "<i4")
    for p in range(calls):
        oimg, ocount, _, _ = osc.render_step(frame, W, H, step, mpl, osc.plane(), O.RNG_HASH, seed=seed, pass_base=p, count=ocount)
    assert img.tobytes() == oimg.tobytes() and np.array_equal(count, ocount) and int(count.max()) == 3 * calls


def _write_cornell_obj(path):
    """An .obj written from the committed mesh arrays (the GPU box has no reference tree)."""
    g = O.load_golden("cornell_obj")
    with open(path, "w") as f:
        for v in g["verts"]:
            f.write("v %r %r %r\n" % (float(v[0]), float(v[1]), float(v[2])))
        for a, b, c in g["faces"]:
            f.write("f %d %d %d\n" % (a + 1, b + 1, c + 1))


@pytest.mark.gpu
def test_facade_render_and_trace_on_gpu(driver, tmp_path):
    obj = str(tmp_path / "cornell_like.obj")
    _write_cornell_obj(obj)
    W, H, passes, mpl, seed = 96, 64, 3, 6, 5
    out = str(tmp_path / "img.f32")
    r = subprocess.run([driver, "render", "obj", obj, str(W), str(H), "1", str(passes), str(mpl), str(seed), out],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(out, "<f4")
    img = raw[: 3 * W * H].reshape(H, W, 3)
    count = raw[3 * W * H:].view("<i4").reshape(H, W)
    assert np.all(count == passes)
    # the same scene through the C ABI / oracle: no usemtl in the written file -> material id -1 on every face
    g = O.load_golden("cornell_obj")
    verts = g["verts"].astype(np.float64)
    mats = np.full(len(g["faces"]), 0xFFFFFFFF, "u4")
    # normals as the reader computes them for a file without vn: normalize(cross(v2-v0, v1-v0))
    osc = O.OracleScene(verts, g["faces"], mats, g["normals"], None)
    frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    oimg, _, _, _ = osc.render(frame, W, H, mpl, passes, osc.plane(), O.RNG_HASH, seed=seed)
    assert img.tobytes() == oimg.tobytes()
    # single-ray Scene::Trace through the facade vs the golden records (geometry is identical)
    t = O.load_golden("trace_cornell_obj")
    rays = t["rays"][:200]
    rp, op = str(tmp_path / "rays.bin"), str(tmp_path / "hits.bin")
    rays.tofile(rp)
    r = subprocess.run([driver, "trace", "obj", obj, rp, op], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    rec = np.fromfile(op, np.dtype([("hit", "<u4"), ("faceID", "<u4"), ("t", "<f8"), ("u", "<f8"), ("v", "<f8"),
                                    ("normal", "<f8", 3)]))
    ref = t["hits"][:200]
    assert np.array_equal(rec["hit"], ref["hit"])
    h = ref["hit"] == 1
    for f in ("faceID", "t", "u", "v", "normal"):
        assert rec[f][h].tobytes() == ref[f][h].tobytes(), f
    # the same rays from four host threads at once (the reference calls Scene::Trace from every OpenMP thread)
    op2 = str(tmp_path / "hits_mt.bin")
    r = subprocess.run([driver, "trace_mt", "obj", obj, rp, op2], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(op2, "rb").read() == open(op, "rb").read()
    # ... and what the resident server and the submission queue of mgpu_trace buy those callers (include/mgpu.h): the same 2 000
    # single-ray calls from 1, 4 and 16 threads through the server (default), through the queue (MGPU_TRACE_SERVER=0) and with a
    # launch per call behind a mutex (MGPU_TRACE_SERVER=0 MGPU_TRACE_QUEUE=0); records identical every time, calls/s in the log
    import re
    rays2 = np.tile(t["rays"][:500], (4, 1))
    rp2 = str(tmp_path / "rays2.bin")
    rays2.tofile(rp2)
    modes = {"server": {}, "queue": {"MGPU_TRACE_SERVER": "0"}, "launch": {"MGPU_TRACE_SERVER": "0", "MGPU_TRACE_QUEUE": "0"}}
    rate, per_launch, ref_bytes = {}, {}, None
    for mode, env in modes.items():
        for nt in (1, 4, 16):
            opq = str(tmp_path / ("hits_%s_%d.bin" % (mode, nt)))
            r = subprocess.run([driver, "trace_mt", "obj", obj, rp2, opq, str(nt)], capture_output=True, text=True, cwd=str(tmp_path),
                               env=dict(os.environ, **env))
            assert r.returncode == 0, r.stdout + r.stderr
            rate[(mode, nt)] = float(re.search(r"([0-9.]+) calls/s", r.stdout).group(1))
            print("%s: %s" % (mode, " | ".join(l for l in r.stdout.splitlines() if l.startswith("trace_mt"))))
            assert ("resident server" in r.stdout) == (mode == "server"), r.stdout
            m = re.search(r"(?:submission queue|resident server): (\d+) calls in (\d+) launches", r.stdout)
            if m:
                per_launch[(mode, nt)] = int(m.group(1)) / max(1, int(m.group(2)))
            b = open(opq, "rb").read()
            ref_bytes = ref_bytes or b
            assert b == ref_bytes, (mode, nt)
    # calls/s are logged, not asserted (wall-clock ratios of 2 000-call runs on a shared box do not belong in a
    # correctness suite; tools/perf_trace_calls.sh and bench.py's scene_trace_one_ray_calls are where they are measured).  What IS
    # asserted is the mechanism: the server serves (nearly) all calls of a run from a handful of launches, and the queue combines
    # concurrent callers into shared launches
    print("Scene::Trace calls/s at 1 / 4 / 16 threads: " + " | ".join(
        "%s %.0f / %.0f / %.0f" % (m, rate[(m, 1)], rate[(m, 4)], rate[(m, 16)]) for m in modes))
    for nt in (1, 4, 16):
        assert per_launch[("server", nt)] >= 100, per_launch   # 2 000 calls: at most 20 server launches (idle exits on a slow host)
    assert per_launch[("queue", 16)] > 1.5 and per_launch[("queue", 1)] == 1.0, per_launch


@pytest.mark.gpu
def test_facade_render_through_the_multi_gpu_frame(driver, tmp_path):
    """mallie::Render / RenderPasses through the multi-GPU frame object (what MALLIE_GPUS=n selects), forced on the one GPU
    there is: strips, ncclSend / ncclRecv to their final rows, read-back -- the image must be the single-GPU image (the driver
    itself checks RenderPasses against Render + AccumImage, this test checks it against the oracle)."""
    obj = str(tmp_path / "cornell_like.obj")
    _write_cornell_obj(obj)
    W, H, passes, mpl, seed = 96, 61, 3, 6, 5
    out = str(tmp_path / "img.f32")
    r = subprocess.run([driver, "render", "obj", obj, str(W), str(H), "1", str(passes), str(mpl), str(seed), out],
                       capture_output=True, text=True, cwd=str(tmp_path), env=dict(os.environ, MGPU_FRAME_FORCE_EXCHANGE="1", MALLIE_GPUS="8"))
    assert r.returncode == 0 and "Render on 1 GPUs" in r.stdout, r.stdout + r.stderr
    raw = np.fromfile(out, "<f4")
    img, count = raw[: 3 * W * H].reshape(H, W, 3), raw[3 * W * H:].view("<i4").reshape(H, W)
    g = O.load_golden("cornell_obj")
    osc = O.OracleScene(g["verts"].astype(np.float64), g["faces"], np.full(len(g["faces"]), 0xFFFFFFFF, "u4"), g["normals"], None)
    frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    oimg, _, _, _ = osc.render(frame, W, H, mpl, passes, osc.plane(), O.RNG_HASH, seed=seed)
    assert img.tobytes() == oimg.tobytes() and np.all(count == passes)


@pytest.mark.gpu
def test_facade_render_on_four_ranks_sharing_the_gpu(driver, tmp_path):
    """MALLIE_GPUS=4 on a one-GPU box (MGPU_FRAME_TRANSPORT=copy: four ranks share the device, copies in place of the RCCL
    pairs): mallie::Render / RenderPasses through a four-rank frame object -- replicas of the scene, strips, staging, placement,
    read-back -- must give the oracle's image."""
    obj = str(tmp_path / "cornell_like.obj")
    _write_cornell_obj(obj)
    W, H, passes, mpl, seed = 96, 61, 3, 6, 5
    out = str(tmp_path / "img.f32")
    r = subprocess.run([driver, "render", "obj", obj, str(W), str(H), "1", str(passes), str(mpl), str(seed), out],
                       capture_output=True, text=True, cwd=str(tmp_path), env=dict(os.environ, MGPU_FRAME_TRANSPORT="copy", MALLIE_GPUS="4"))
    assert r.returncode == 0 and "Render on 4 GPUs" in r.stdout, r.stdout + r.stderr
    raw = np.fromfile(out, "<f4")
    img, count = raw[: 3 * W * H].reshape(H, W, 3), raw[3 * W * H:].view("<i4").reshape(H, W)
    g = O.load_golden("cornell_obj")
    osc = O.OracleScene(g["verts"].astype(np.float64), g["faces"], np.full(len(g["faces"]), 0xFFFFFFFF, "u4"), g["normals"], None)
    frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    oimg, _, _, _ = osc.render(frame, W, H, mpl, passes, osc.plane(), O.RNG_HASH, seed=seed)
    assert img.tobytes() == oimg.tobytes() and np.all(count == passes)


@pytest.mark.gpu
def test_facade_render_in_the_fast_mode(driver, tmp_path):
    """MALLIE_FAST=1: mallie::Render / RenderPasses in fp32 (plain and through the multi-GPU frame): close to the oracle's
    frame -- rms per-pixel L2 of the pixel means <= 1e-3 at this tiny size, under 0.5 % of the pixels moved by more than
    1e-3 -- and not equal to it (the switch did something); the driver's own check RenderPasses == Render + AccumImage holds
    in the fast mode too."""
    obj = str(tmp_path / "cornell_like.obj")
    _write_cornell_obj(obj)
    W, H, passes, mpl, seed = 96, 64, 3, 6, 5
    g = O.load_golden("cornell_obj")
    osc = O.OracleScene(g["verts"].astype(np.float64), g["faces"], np.full(len(g["faces"]), 0xFFFFFFFF, "u4"), g["normals"], None)
    frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    oimg, _, _, _ = osc.render(frame, W, H, mpl, passes, osc.plane(), O.RNG_HASH, seed=seed)
    imgs = []
    for extra in ({}, {"MGPU_FRAME_FORCE_EXCHANGE": "1"}):
        out = str(tmp_path / "img.f32")
        r = subprocess.run([driver, "render", "obj", obj, str(W), str(H), "1", str(passes), str(mpl), str(seed), out],
                           capture_output=True, text=True, cwd=str(tmp_path), env=dict(os.environ, MALLIE_FAST="1", **extra))
        assert r.returncode == 0, r.stdout + r.stderr
        raw = np.fromfile(out, "<f4")
        img, count = raw[: 3 * W * H].reshape(H, W, 3), raw[3 * W * H:].view("<i4").reshape(H, W)
        assert np.all(count == passes) and img.tobytes() != oimg.tobytes()
        l2 = np.sqrt((((img.astype(np.float64) - oimg) / passes) ** 2).sum(-1))
        assert np.sqrt((l2 ** 2).mean()) <= 1e-3 and (l2 > 1e-3).mean() <= 5e-3, (np.sqrt((l2 ** 2).mean()), (l2 > 1e-3).mean())
        imgs.append(img)
    assert imgs[0].tobytes() == imgs[1].tobytes()


@pytest.mark.gpu
def test_facade_render_in_the_reference_stream(driver, tmp_path):
    """MALLIE_RNG_STREAM=1: mallie::Render draws from the reference's own serial random stream, continued from call to call
    (three Render() calls + AccumImage in the driver) -- the oracle's run of three passes in that stream, bit for bit; the
    driver's own check (RenderPasses == Render + AccumImage) is skipped by the stream moving on, so only the sum is compared."""
    obj = str(tmp_path / "cornell_like.obj")
    _write_cornell_obj(obj)
    W, H, passes, mpl = 96, 64, 3, 16
    out = str(tmp_path / "img.f32")
    r = subprocess.run([driver, "render", "obj", obj, str(W), str(H), "1", str(passes), str(mpl), "1", out],
                       capture_output=True, text=True, cwd=str(tmp_path), env=dict(os.environ, MALLIE_RNG_STREAM="1"))
    assert r.returncode in (0, 7), r.stdout + r.stderr   # 7: the driver's second rendering continued the stream, as it must
    img = np.fromfile(out, "<f4")[: 3 * W * H].reshape(H, W, 3)
    g = O.load_golden("cornell_obj")
    osc = O.OracleScene(g["verts"].astype(np.float64), g["faces"], np.full(len(g["faces"]), 0xFFFFFFFF, "u4"), g["normals"], None)
    frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    oimg, _, _, _ = osc.render(frame, W, H, mpl, passes, osc.plane(), O.RNG_STREAM, stream_state=np.array(O.REFERENCE_SEED, "<u4"))
    assert img.tobytes() == oimg.tobytes()


@pytest.mark.gpu
def test_facade_renders_a_vox_scene_with_its_palette(driver, tmp_path):
    """Scene::Init(.vox) -> mallie::Render on the GPU: the palette materials colour the paths (R, G, B differ) and the
    image equals the oracle's for the same arrays, materials and seeding."""
    vox = os.path.join(ROOT, "tests", "golden", "objs", "tiny_rgba.vox")
    W, H, passes, mpl, seed = 80, 64, 2, 6, 3
    out = str(tmp_path / "img.f32")
    r = subprocess.run([driver, "render", "vox", vox, str(W), str(H), "1", str(passes), str(mpl), str(seed), out],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    img = np.fromfile(out, "<f4")[: 3 * W * H].reshape(H, W, 3)
    g = O.load_golden("objload_vox_rgba")
    osc = O.OracleScene(g["verts"].astype(np.float64), g["faces"], g["matIDs"], None, None, g["nodes"], g["indices"],
                        mat_diffuse=g["materials"])
    frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    oimg, _, _, _ = osc.render(frame, W, H, mpl, passes, osc.plane(), O.RNG_HASH, seed=seed)
    assert img.tobytes() == oimg.tobytes(), "%d pixels differ" % int((img != oimg).any(-1).sum())
    assert not np.array_equal(img[..., 0], img[..., 2])


@pytest.mark.gpu
def test_facade_render_panoramic_on_gpu(driver, tmp_path):
    """mallie::RenderPanoramic through the Mallie-named facade (the console driver's call) vs the oracle, same seeding."""
    obj = str(tmp_path / "cornell_like.obj")
    _write_cornell_obj(obj)
    W, H, seed = 96, 48, 7
    g = O.load_golden("cornell_obj")
    osc = O.OracleScene(g["verts"].astype(np.float64), g["faces"], np.full(len(g["faces"]), 0xFFFFFFFF, "u4"), g["normals"], None)
    origin = O.camera_frame((0.0, 1.0, 4.0), (0, 0, 0), width=W, height=H)[:3]
    for stereo in (0, 1):
        out = str(tmp_path / ("pano%d.f32" % stereo))
        r = subprocess.run([driver, "panoramic", "obj", obj, str(W), str(H), str(stereo), str(seed), out],
                           capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 0, r.stdout + r.stderr
        raw = np.fromfile(out, "<f4")
        img = raw[: 3 * W * H].reshape(H, W, 3)
        count = raw[3 * W * H:].view("<i4").reshape(H, W)
        assert np.all(count == 10)
        oimg, _, _, _ = osc.render_panoramic(origin, W, H, stereo, 16, 10, O.RNG_HASH, seed=seed)
        assert img.tobytes() == oimg.tobytes(), "%d pixels differ" % int((img != oimg).any(-1).sum())


@pytest.mark.gpu
def test_facade_build_routes_large_meshes_to_the_device_builder(driver, tmp_path):
    """BVHAccel::Build sends meshes of >= 65536 triangles to mgpu_bvh_build_device; the tree must equal the host
    builder's (and MALLIE_BVH_BUILD=host must give the same file)."""
    from mallie_amd.scenes import suzanne_grid
    c = O.load_golden("cornell_obj")
    verts, faces, _, _ = suzanne_grid(c["verts"], c["faces"], 9)          # 78 408 triangles
    assert len(faces) >= 65536
    obj = str(tmp_path / "grid9.obj")
    with open(obj, "w") as f:
        for v in verts:
            f.write("v %r %r %r\n" % (float(v[0]), float(v[1]), float(v[2])))
        for a, b, cc in faces:
            f.write("f %d %d %d\n" % (a + 1, b + 1, cc + 1))
    out = {}
    for mode in ("device", "host"):
        prefix = str(tmp_path / ("m_" + mode))
        env = dict(os.environ, MALLIE_BVH_BUILD=mode)
        r = subprocess.run([driver, "mesh", "obj", obj, "1.0", prefix], capture_output=True, text=True, env=env,
                           cwd=str(tmp_path))
        assert r.returncode == 0, r.stdout + r.stderr
        out[mode] = read_mesh(prefix)
    assert np.array_equal(out["device"]["verts"], verts) and np.array_equal(out["device"]["faces"], faces)
    nodes, idx, _ = M.bvh_build(verts, faces)
    for mode in ("device", "host"):
        assert out[mode]["nodes"].tobytes() == nodes.tobytes() and np.array_equal(out[mode]["indices"], idx), mode


@pytest.mark.gpu
def test_threaded_callers_stress_driver(tmp_path_factory):
    """tests/cpp/stress_driver.cc: 16 threads of one-ray Scene::Trace-shaped callers (mailbox server + submission queue), frames
    rendered beside them, scenes created and destroyed meanwhile; every record must equal the batched call's, every frame the
    first frame's bytes.  (tools/sanitize_gpu.sh runs the same driver against ASan / TSan builds of the library:
    profiles/r5_sanitizers.txt.)"""
    out = str(tmp_path_factory.mktemp("stress") / "stress_driver")
    libdir = os.path.dirname(M.lib_path())
    cmd = ["g++", "-O1", "-std=c++11", "-pthread", os.path.join(ROOT, "tests", "cpp", "stress_driver.cc"), "-L", libdir,
           "-lmallie_mgpu", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([out, "16", "400", "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 records differ" in r.stdout and "0 failed calls" in r.stdout, r.stdout + r.stderr
