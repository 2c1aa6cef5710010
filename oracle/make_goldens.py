#!/usr/bin/env python3
"""oracle/make_goldens.py -- TEST INFRASTRUCTURE ONLY.

Generates tests/golden/*.npz from the UNMODIFIED reference, compiled from /root/reference by
`make -C oracle ref` into oracle/_ref/ref_driver (see oracle/ref_driver.cc).  Runs only in the build
container (it needs /root/reference for the scene files); the GPU box and the test-suite consume the
committed .npz files.  Nothing here is imported by the product.

Every vector below is an OUTPUT of the reference binary (or an input fed to it):
  camera.npz        Camera::BuildCameraFrame frames + Camera::GenerateRay probes        (camera.cc:40-240)
  cornell_obj.npz   Mesh arrays after Scene::Init(obj) + BVH nodes/indices               (scene.cc:66, bvh_accel.cc:445)
  cornell_eson.npz  same through the ESON loader (no facevarying normals)                (mesh_loader.cc:212)
  teapot_obj.npz    same for teapot.obj
  trace_*.npz       (Ray -> Intersection) batches through Scene::Trace                   (scene.cc:253)
  render_*.npz      Render() images, 1..2 consecutive passes, OMP_NUM_THREADS=1           (render.cc:593)
  pano_*.npz        RenderPanoramic() images (mono / stereo), OMP_NUM_THREADS=1            (render.cc:710)

Rules: CWD=/root/reference so tinyobj finds the .mtl; one process per render config
(Render() keeps `static bool initial_pass`); OMP_NUM_THREADS=1.
"""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MALLIE_REF", "/root/reference")
DRIVER = os.path.join(HERE, "_ref", "ref_driver")
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

NODE_DT = np.dtype([("bmin", "<f8", 3), ("bmax", "<f8", 3), ("flag", "<i4"), ("axis", "<i4"), ("data", "<u4", 2)])
HIT_DT = np.dtype([("hit", "<u4"), ("faceID", "<u4"), ("materialID", "<u4"), ("f0", "<u4"), ("f1", "<u4"),
                   ("f2", "<u4"), ("t", "<f8"), ("u", "<f8"), ("v", "<f8"), ("position", "<f8", 3),
                   ("geometricNormal", "<f8", 3), ("normal", "<f8", 3), ("texcoord", "<f8", 2)])
assert NODE_DT.itemsize == 64 and HIT_DT.itemsize == 136


def run(args, stdin=None, cwd=None):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([DRIVER] + [str(a) for a in args], cwd=cwd or REF, env=env, input=stdin, capture_output=True,
                       text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
        raise SystemExit("ref_driver failed: %r" % (args,))
    return r


def read_mesh(prefix):
    with open(prefix + ".mesh", "rb") as f:
        nv, nf, hn, hu = struct.unpack("<QQBB", f.read(18))
        verts = np.frombuffer(f.read(24 * nv), "<f8").reshape(nv, 3)
        faces = np.frombuffer(f.read(12 * nf), "<u4").reshape(nf, 3)
        mats = np.frombuffer(f.read(4 * nf), "<u4")
        normals = np.frombuffer(f.read(72 * nf), "<f8").reshape(nf, 9) if hn else np.zeros((0, 9))
        uvs = np.frombuffer(f.read(48 * nf), "<f8").reshape(nf, 6) if hu else np.zeros((0, 6))
    with open(prefix + ".bvh", "rb") as f:
        (nn,) = struct.unpack("<Q", f.read(8))
        nodes = np.frombuffer(f.read(64 * nn), NODE_DT).copy()
        (ni,) = struct.unpack("<Q", f.read(8))
        idx = np.frombuffer(f.read(4 * ni), "<u4")
    # the reference never initialises BVHNode::axis of a leaf (bvh_accel.cc:343-360): the dumped value is stack
    # garbage, so it is masked here and ignored by every comparison.
    nodes["axis"][nodes["flag"] == 1] = 0
    return dict(verts=verts, faces=faces, matIDs=mats, normals=normals, uvs=uvs, has_normals=np.uint8(hn),
                has_uvs=np.uint8(hu), nodes=nodes, indices=idx)


def gen_mesh(tmp, kind, fname, name, cwd=None):
    prefix = os.path.join(tmp, name)
    run(["mesh", kind, fname, 1.0, prefix], cwd=cwd)
    m = read_mesh(prefix)
    # uvs of these scenes are all-zero placeholders (mesh_loader.cc:64-65); store only the flag when so.
    if m["uvs"].size and not m["uvs"].any():
        m["uvs"] = np.zeros((0, 6))
        m["uvs_all_zero"] = np.uint8(1)
    else:
        m["uvs_all_zero"] = np.uint8(0)
    # vertex coordinates came through float32 (tinyobj / ESON store float): keep them compact when lossless
    if np.array_equal(m["verts"], m["verts"].astype(np.float32).astype(np.float64)):
        m["verts"] = m["verts"].astype(np.float32)
    if kind == "vox":  # Scene::GetMaterial(0..255).diffuse as the reference's reader filled materials_
        m["materials"] = np.fromfile(prefix + ".mat", "<f8").reshape(256, 3)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **m)
    print(name, "verts", m["verts"].shape, "faces", m["faces"].shape, "nodes", m["nodes"].shape,
          "normals", m["normals"].shape)
    return m


def make_rays(rng, m, n_cam, n_rand, n_vertex, n_axis, eye):
    """A ray set that exercises primary-like rays, incoherent interior rays, exact-vertex / edge hits (t ties
    between neighbouring triangles), axis-parallel directions (1/0 -> inf) and misses."""
    v = m["verts"].astype(np.float64)
    lo, hi = v.min(0), v.max(0)
    rays = []
    # camera-like
    tgt = lo + (hi - lo) * rng.random((n_cam, 3))
    d = tgt - eye
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays.append(np.hstack([np.tile(eye, (n_cam, 1)), d]))
    # interior random
    o = lo + (hi - lo) * (0.1 + 0.8 * rng.random((n_rand, 3)))
    d = rng.normal(size=(n_rand, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays.append(np.hstack([o, d]))
    # aimed exactly at mesh vertices and edge midpoints
    f = m["faces"][rng.integers(0, len(m["faces"]), n_vertex)]
    pick = rng.integers(0, 3, n_vertex)
    tv = v[f[np.arange(n_vertex), pick]]
    mid = 0.5 * (v[f[:, 0]] + v[f[:, 1]])
    tv[::2] = mid[::2]
    o = np.tile(eye, (n_vertex, 1)) + rng.normal(scale=0.5, size=(n_vertex, 3))
    d = tv - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays.append(np.hstack([o, d]))
    # axis-parallel (zero direction components -> +-inf inverse direction), unnormalised lengths too
    o = lo + (hi - lo) * rng.random((n_axis, 3))
    d = np.zeros((n_axis, 3))
    ax = rng.integers(0, 3, n_axis)
    d[np.arange(n_axis), ax] = rng.choice([-1.0, 1.0, 2.5, -0.25], n_axis)
    second = rng.random(n_axis) < 0.3
    d[second, (ax[second] + 1) % 3] = rng.normal(size=second.sum())
    rays.append(np.hstack([o, d]))
    return np.ascontiguousarray(np.vstack(rays), dtype="<f8")


def gen_trace(tmp, kind, fname, name, rays):
    rp = os.path.join(tmp, name + ".rays")
    op = os.path.join(tmp, name + ".hits")
    rays.tofile(rp)
    run(["trace", kind, fname, 1.0, rp, op])
    hits = np.fromfile(op, HIT_DT)
    assert len(hits) == len(rays)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), rays=rays, hits=hits)
    print(name, len(rays), "rays", int(hits["hit"].sum()), "hits")


def gen_render(tmp, kind, fname, name, W, H, plane, passes, eye, lookat, up=(0, 1, 0), quat=(0, 0, 0, 0), store=True,
               step=1):
    prefix = os.path.join(tmp, name)
    run(["render", kind, fname, 1.0, W, H, int(plane), passes, *eye, *lookat, *up, *quat, prefix] + ([step] if step != 1 else []))
    imgs = [np.fromfile("%s.pass%d.f32" % (prefix, p), "<f4").reshape(H, W, 3) for p in range(passes)]
    count = np.fromfile(prefix + ".count.i32", "<i4").reshape(H, W)
    meta = dict(W=W, H=H, plane=int(plane), passes=passes, eye=np.array(eye, "f8"), lookat=np.array(lookat, "f8"),
                up=np.array(up, "f8"), quat=np.array(quat, "f8"), maxPathLength=16, scene=name.split("_")[1])
    if step != 1:
        meta["step"] = step
    if store:
        np.savez_compressed(os.path.join(OUT, name + ".npz"), images=np.stack(imgs), count=count, **meta)
    else:
        # large frame: keep scalars + a digest only
        im = imgs[0]
        np.savez_compressed(os.path.join(OUT, name + ".npz"),
                            sha256=np.frombuffer(hashlib.sha256(im.tobytes()).digest(), "u1"),
                            mean_r=np.float64(im[..., 0].astype(np.float64).mean()),
                            nonzero=np.int64((im[..., 0] != 0).sum()), rows=im[::64].copy(),
                            row_ids=np.arange(0, H, 64), count_min=count.min(), count_max=count.max(), **meta)
    print(name, [float(i[..., 0].mean()) for i in imgs])


def gen_pano(tmp, kind, fname, name, W, H, stereo, eye, lookat=(0, 0, 0), up=(0, 1, 0), quat=(0, 0, 0, 0)):
    """One RenderPanoramic() call (render.cc:710-763; 10 PathTraceEnv samples per pixel, maxPathLength 16)."""
    prefix = os.path.join(tmp, name)
    run(["panoramic", kind, fname, 1.0, W, H, int(stereo), *eye, *lookat, *up, *quat, prefix])
    img = np.fromfile(prefix + ".f32", "<f4").reshape(H, W, 3)
    count = np.fromfile(prefix + ".count.i32", "<i4").reshape(H, W)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), image=img, count=count, W=W, H=H, stereo=int(stereo),
                        eye=np.array(eye, "f8"), lookat=np.array(lookat, "f8"), up=np.array(up, "f8"),
                        quat=np.array(quat, "f8"), maxPathLength=16, samples=10, scene=name.split("_")[1])
    print(name, float(img[..., 0].mean()), int((img[..., 0] != 0).sum()), "non-zero pixels")


def gen_pano_all(tmp):
    # main_console.cc:104-106 puts the eye on a circle of radius 4 at height 1 (frame 0: (0, 1, 4)) and asks for stereo
    gen_pano(tmp, "obj", "cornellbox_suzanne.obj", "pano_cornell_stereo_96x64", 96, 64, True, (0.0, 1.0, 4.0))
    gen_pano(tmp, "obj", "cornellbox_suzanne.obj", "pano_cornell_mono_80x40", 80, 40, False, (0.0, 1.0, 4.0))
    gen_pano(tmp, "obj", "cornellbox_suzanne.obj", "pano_cornell_stereo_50x37_view2", 50, 37, True, (2.5, 3.0, -1.5),
             quat=(0.05, -0.1, 0.02, 0.99))
    gen_pano(tmp, "obj", "teapot.obj", "pano_teapot_mono_64x32", 64, 32, False, (0.0, 60.0, 120.0))


def gen_step_all(tmp):
    gen_render(tmp, "obj", "cornellbox_suzanne.obj", "render_cornell_obj_64_plane_step2_2pass", 64, 64, True, 2, (0, 0, 20), (0, 0, 0),
               step=2)
    gen_render(tmp, "obj", "cornellbox_suzanne.obj", "render_cornell_obj_60x48_noplane_step4", 60, 48, False, 1, (0, 0, 20), (0, 0, 0),
               step=4)


def gen_aov(tmp):
    """ShowNormal / ShowUV (render.cc:458-516) through oracle/_ref/ref_aov_driver (which #includes the unmodified render.cc)."""
    drv = os.path.join(HERE, "_ref", "ref_aov_driver")
    for name, fname, W, H, eye, la, mode in [("aov_cornell_normal_64x48", "cornellbox_suzanne.obj", 64, 48, (0, 0, 20), (0, 0, 0), "normal"),
                                             ("aov_teapot_normal_72x40", "teapot.obj", 72, 40, (0, 40, 250), (0, 40, 0), "normal"),
                                             ("aov_teapot_uv_64x48", "teapot.obj", 64, 48, (0, 40, 250), (0, 40, 0), "uv"),
                                             ("aov_cornell_uv_32x24", "cornellbox_suzanne.obj", 32, 24, (0, 0, 20), (0, 0, 0), "uv")]:
        out = os.path.join(tmp, name + ".f32")
        p = subprocess.run([drv, "obj", fname, str(W), str(H), *map(str, eye), *map(str, la), mode, out], cwd=REF,
                           env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        img = np.fromfile(out, "<f4").reshape(H, W, 3)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), image=img, W=W, H=H, eye=np.array(eye, "f8"), lookat=np.array(la, "f8"),
                            mode=0 if mode == "normal" else 1, scene=name.split("_")[1])
        print(name, float(img.mean()), int((img != 0).any(-1).sum()), "non-black pixels")


def gen_boundary(tmp):
    """Camera::GenerateEnvRay / GenerateStereoEnvRay (camera.cc:242-329) and Plane::intersect (prim-plane.cc:8-44) probes."""
    rng = np.random.default_rng(11)
    W, H = 96, 64
    eye, la = (0.5, 1.0, 4.0), (0.0, 0.5, 0.0)
    uv = np.column_stack([rng.random(64) * W, rng.random(64) * H])
    uv[:6] = [(0, 0), (W - 1, H - 1), (W / 2, H / 2), (W / 2, H / 2 - 1), (17, 31.999), (17, 32)]
    out = os.path.join(tmp, "env.bin")
    p = subprocess.run([DRIVER, "envrays", str(W), str(H), "45", *map(str, eye), *map(str, la), "0", "1", "0", "0", "0", "0", "0", out],
                       input="\n".join("%r %r" % (float(u), float(v)) for u, v in uv), text=True, cwd=REF, capture_output=True)
    assert p.returncode == 0, p.stderr
    env = np.fromfile(out, "<f8").reshape(-1, 12)
    pl = (0.25, 1.0, -0.125, 0.75)
    rays = np.zeros((96, 7))
    rays[:, 0:3] = rng.normal(size=(96, 3)) * 3
    rays[:, 3:6] = rng.normal(size=(96, 3))
    rays[:, 6] = np.where(rng.random(96) < 0.5, 1.0e30, rng.random(96) * 6)
    rays[0, 3:6] = (1, -0.25, 0.0)          # parallel to the plane: |v.n| below the epsilon
    rays[1, 3:6] = 0                        # zero direction: normalize() leaves it alone
    rays[2, 3:6] *= 1e-9                    # shorter than normalize()'s 1e-6 guard
    rp, op = os.path.join(tmp, "pl.rays"), os.path.join(tmp, "pl.out")
    rays.tofile(rp)
    run(["plane", *pl, rp, op])
    np.savez_compressed(os.path.join(OUT, "boundary.npz"), W=W, H=H, eye=np.array(eye), lookat=np.array(la), uv=uv, env=env,
                        plane=np.array(pl), plane_rays=rays, plane_out=np.fromfile(op, "<f8").reshape(-1, 22))
    print("boundary", env.shape, int(np.fromfile(op, "<f8").reshape(-1, 22)[:, 0].sum()), "plane hits")


def gen_camera(tmp):
    rng = np.random.default_rng(7)
    cfgs = [
        (512, 512, 45.0, (0, 0, 20), (0, 0, 0), (0, 1, 0), (0, 0, 0, 0)),
        (1920, 1080, 45.0, (0, 0, 20), (0, 0, 0), (0, 1, 0), (0, 0, 0, 0)),
        (1920, 1080, 45.0, (0, 40, 250), (0, 40, 0), (0, 1, 0), (0, 0, 0, 0)),
        (1920, 1080, 45.0, (0, 40, 80), (0, 0, 0), (0, 1, 0), (0, 0, 0, 0)),
        (640, 480, 60.0, (3.5, 2.25, -7.75), (0.5, 1.0, 0.25), (0, 1, 0), (0, 0, 0, 0)),
        (512, 512, 45.0, (0, 0, 20), (0, 0, 0), (0, 1, 0), (6.123233995736766e-17, 0, 0, 1)),  # main_sdl.cc:593
        (333, 777, 30.0, (1, 2, 3), (-4, 0.5, -6), (0.1, 0.9, 0.2), (0.1, -0.2, 0.3, 0.9)),
        (800, 600, 75.0, (-12, 6, 9), (0, 1, 0), (0, 1, 0), (0.0, 0.38268343236508978, 0.0, 0.92387953251128674)),
    ]
    frames, probes, rays = [], [], []
    for i, (W, H, fov, eye, la, up, q) in enumerate(cfgs):
        uv = np.column_stack([rng.random(16) * W, rng.random(16) * H])
        uv[0] = (0.0, 0.0)
        uv[1] = (W - 0.5, H - 0.5)
        # px + float jitter is a float32 value in the reference (render.cc:387-391): probe such values too
        uv[2:8] = uv[2:8].astype(np.float32)
        out = os.path.join(tmp, "cam%d.bin" % i)
        run(["camera", W, H, repr(fov), *map(repr, map(float, eye)), *map(repr, map(float, la)),
             *map(repr, map(float, up)), *map(repr, map(float, q)), out],
            stdin="".join("%r %r\n" % (float(a), float(b)) for a, b in uv))
        d = np.fromfile(out, "<f8")
        frames.append(d[:12])
        rays.append(d[12:].reshape(-1, 6))
        probes.append(uv)
    cfg_arr = np.array([[W, H, fov, *eye, *la, *up, *q] for (W, H, fov, eye, la, up, q) in cfgs], "f8")
    np.savez_compressed(os.path.join(OUT, "camera.npz"), cfg=cfg_arr, frames=np.array(frames), probes=np.array(probes),
                        rays=np.array(rays))
    print("camera", len(cfgs), "frames; corner[0] =", frames[0][3:6], " corner[1] =", frames[1][3:6])


def main():
    if not os.path.exists(DRIVER):
        raise SystemExit("build the reference driver first: make -C oracle ref")
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["vox"]:  # only the MagicaVoxel reader goldens (authored inputs: tests/golden/objs/make_vox.py)
        with tempfile.TemporaryDirectory() as tmp:
            objs = os.path.join(OUT, "objs")
            gen_mesh(tmp, "vox", "tiny.vox", "objload_vox_default", cwd=objs)
            gen_mesh(tmp, "vox", "tiny_rgba.vox", "objload_vox_rgba", cwd=objs)
        return
    if sys.argv[1:] == ["step"]:  # only the Render(step > 1) goldens (render.cc:684-696)
        with tempfile.TemporaryDirectory() as tmp:
            gen_step_all(tmp)
        return
    if sys.argv[1:] == ["aov"]:
        with tempfile.TemporaryDirectory() as tmp:
            gen_aov(tmp)
        return
    if sys.argv[1:] == ["boundary"]:
        with tempfile.TemporaryDirectory() as tmp:
            gen_boundary(tmp)
        return
    if sys.argv[1:] == ["pano"]:  # only the RenderPanoramic goldens (keeps the other fixtures' bytes untouched)
        with tempfile.TemporaryDirectory() as tmp:
            gen_pano_all(tmp)
        return
    with tempfile.TemporaryDirectory() as tmp:
        gen_pano_all(tmp)
        gen_step_all(tmp)
        gen_boundary(tmp)
        gen_aov(tmp)
        gen_camera(tmp)
        mc = gen_mesh(tmp, "obj", "cornellbox_suzanne.obj", "cornell_obj")
        gen_mesh(tmp, "eson", "cornellbox_suzanne.eson", "cornell_eson")
        mt = gen_mesh(tmp, "obj", "teapot.obj", "teapot_obj")
        # authored .obj inputs (tests/golden/objs/) through the reference loader: loader-behaviour goldens for mesh_io.cc
        objs = os.path.join(OUT, "objs")
        gen_mesh(tmp, "obj", "quirks.obj", "objload_quirks", cwd=objs)
        gen_mesh(tmp, "obj", "nomtl.obj", "objload_nomtl", cwd=objs)
        gen_mesh(tmp, "vox", "tiny.vox", "objload_vox_default", cwd=objs)
        gen_mesh(tmp, "vox", "tiny_rgba.vox", "objload_vox_rgba", cwd=objs)
        rng = np.random.default_rng(20260929)
        gen_trace(tmp, "obj", "cornellbox_suzanne.obj", "trace_cornell_obj",
                  make_rays(rng, mc, 1500, 1500, 700, 300, np.array([0.0, 0.0, 20.0])))
        gen_trace(tmp, "eson", "cornellbox_suzanne.eson", "trace_cornell_eson",
                  make_rays(rng, mc, 300, 300, 100, 50, np.array([0.0, 5.0, 20.0])))
        gen_trace(tmp, "obj", "teapot.obj", "trace_teapot_obj",
                  make_rays(rng, mt, 1200, 800, 400, 100, np.array([0.0, 40.0, 250.0])))
        eye, la = (0, 0, 20), (0, 0, 0)
        gen_render(tmp, "obj", "cornellbox_suzanne.obj", "render_cornell_obj_64_plane_2pass", 64, 64, True, 2, eye, la)
        gen_render(tmp, "obj", "cornellbox_suzanne.obj", "render_cornell_obj_64_noplane", 64, 64, False, 1, eye, la)
        gen_render(tmp, "obj", "cornellbox_suzanne.obj", "render_cornell_obj_128x96_plane", 128, 96, True, 1, eye, la)
        gen_render(tmp, "eson", "cornellbox_suzanne.eson", "render_cornell_eson_48_plane", 48, 48, True, 1, eye, la)
        gen_render(tmp, "obj", "cornellbox_suzanne.obj", "render_cornell_obj_40x56_view2", 40, 56, True, 1,
                   (6.5, 7.0, 14.0), (0.0, 3.0, 0.0), quat=(0.05, -0.1, 0.02, 0.99))
        gen_render(tmp, "obj", "teapot.obj", "render_teapot_obj_64x48_plane", 64, 48, True, 1, (0, 40, 250), (0, 40, 0))
        gen_render(tmp, "obj", "cornellbox_suzanne.obj", "render_cornell_obj_512_plane_digest", 512, 512, True, 1, eye, la,
                   store=False)


if __name__ == "__main__":
    main()
