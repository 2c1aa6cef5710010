"""The BASELINE.json configurations as data: scene arrays, camera, path length, passes.

Meshes come from the committed fixtures (tests/golden/*.npz: the mesh arrays the reference's loaders produce for
cornellbox_suzanne.obj and teapot.obj) and from the suzanne-grid generator; there is no dataset to download."""
import os

import numpy as np

from . import mgpu
from .scenes import suzanne_grid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = {
    "c2": dict(name="C2", scene="cornellbox_suzanne.obj", mesh="cornell_obj", width=1920, height=1080, spp=16, bounces=4,
               eye=(0.0, 0.0, 20.0), lookat=(0.0, 0.0, 0.0), plane=True, seed=1),
    "c3": dict(name="C3", scene="teapot.obj", mesh="teapot_obj", width=1920, height=1080, spp=64, bounces=8,
               eye=(0.0, 40.0, 250.0), lookat=(0.0, 40.0, 0.0), plane=True, seed=1),
    "c4": dict(name="C4", scene="suzanne grid 32x32", grid=32, width=1920, height=1080, spp=16, bounces=4,
               eye=(0.0, 40.0, 80.0), lookat=(0.0, 0.0, 0.0), plane=True, seed=1),
    "c5": dict(name="C5", scene="suzanne grid 102x102", grid=102, width=3840, height=2160, spp=64, bounces=4,
               eye=(0.0, 40.0, 80.0), lookat=(0.0, 0.0, 0.0), plane=True, seed=1),
}


def _golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)


def mesh_arrays(cfg):
    """(verts float64 [V,3], faces uint32 [F,3], matIDs uint32 [F], facevarying normals float64 [F,9])."""
    if "grid" in cfg:
        c = _golden("cornell_obj")
        return suzanne_grid(c["verts"], c["faces"], cfg["grid"])
    g = _golden(cfg["mesh"])
    return g["verts"].astype(np.float64), g["faces"], g["matIDs"], g["normals"]


def n_tris(cfg):
    """Triangles of a configuration's mesh (without building it)."""
    if "grid" in cfg:
        return cfg["grid"] ** 2 * 968
    return int(len(_golden(cfg["mesh"])["faces"]))


def make_scene(cfg, device=0, min_leaf=None):
    """Scene of a configuration on `device`: BVH by this library's builder (device builder from 65 536 triangles up).
    min_leaf: BVHBuildOptions::minLeafPrimitives (bvh_accel.h:33-43) when not the reference's default of 16."""
    verts, faces, mats, normals = mesh_arrays(cfg)
    nodes = idx = None
    if min_leaf is not None:
        nodes, idx, _ = mgpu.bvh_build(verts, faces, minLeaf=int(min_leaf), device=device if len(faces) >= 65536 else None)
    elif len(faces) >= 65536:
        nodes, idx, _ = mgpu.bvh_build(verts, faces, device=device)
    return mgpu.Scene(verts, faces, mats, normals, None, nodes, idx, device=device)


def camera(cfg):
    return mgpu.camera_frame(cfg["eye"], cfg["lookat"], width=cfg["width"], height=cfg["height"])


def describe(cfg, nf):
    return "%s (%d tris), %dx%d, %d spp, %d bounces (maxPathLength %d), plane %s, eye %s, per-(pixel,pass) xorshift128 seeding, seed %d" % (
        cfg["scene"], nf, cfg["width"], cfg["height"], cfg["spp"], cfg["bounces"], cfg["bounces"] + 1,
        "on" if cfg["plane"] else "off", tuple(cfg["eye"]), cfg["seed"])
