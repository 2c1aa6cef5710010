"""Pins the oracle (oracle/mallie_oracle.c) bit-for-bit to vectors produced by the UNMODIFIED reference binary
(oracle/make_goldens.py).  CPU only.  If this file is green the oracle may be trusted as the checker."""
import hashlib

import numpy as np
import pytest

import oracle_lib as O

MESHES = ["cornell_obj", "cornell_eson", "teapot_obj"]


def test_camera_frames_bit_exact():
    g = O.load_golden("camera")
    for cfg, frame, probes, rays in zip(g["cfg"], g["frames"], g["probes"], g["rays"]):
        W, H, fov = int(cfg[0]), int(cfg[1]), cfg[2]
        f = O.camera_frame(cfg[3:6], cfg[6:9], cfg[9:12], cfg[12:16], fov, W, H)
        assert f.tobytes() == frame.tobytes(), (cfg, f, frame)
        for (u, v), r in zip(probes, rays):
            assert O.generate_ray(f, u, v).tobytes() == r.tobytes()


def test_camera_survey_values():
    # SURVEY.md F9 scalars, measured from the reference
    f = O.camera_frame((0, 0, 20), (0, 0, 0), width=512, height=512)
    assert tuple(f[3:6]) == (-256.0, 256.0, -598.03866361323981)
    f = O.camera_frame((0, 0, 20), (0, 0, 0), width=1920, height=1080)
    assert tuple(f[3:6]) == (-960.0, 540.0, -1283.6753060591777)
    assert tuple(f[6:9]) == (1.0, -0.0, 0.0) and tuple(f[9:12]) == (0.0, -1.0, -0.0)


@pytest.mark.parametrize("name", MESHES)
def test_bvh_build_identical(name):
    g = O.load_golden(name)
    nodes, idx, stats = O.bvh_build(g["verts"], g["faces"])
    assert len(nodes) == len(g["nodes"])
    assert np.array_equal(idx, g["indices"])
    for field in ("bmin", "bmax", "flag", "axis", "data"):
        assert nodes[field].tobytes() == g["nodes"][field].tobytes(), field
    assert stats["numLeafNodes"] + stats["numBranchNodes"] == len(nodes)


def test_bvh_survey_values():
    g = O.load_golden("cornell_obj")
    n0 = g["nodes"][0]
    assert len(g["nodes"]) == 205 and n0["flag"] == 0 and n0["axis"] == 1 and tuple(n0["data"]) == (1, 204)
    assert tuple(n0["bmin"]) == (-5.144927024841536, -0.031672999263037127, -4.9552760124208817)
    assert tuple(n0["bmax"]) == (4.658843040466536, 9.7728328704836258, 4.7864861488344559)
    _, _, st = O.bvh_build(g["verts"], g["faces"])
    assert st == dict(maxTreeDepth=13, numLeafNodes=103, numBranchNodes=102)


@pytest.mark.parametrize("name", MESHES)
def test_trace_bit_exact(name):
    t = O.load_golden("trace_" + name)
    for own in (False, True):
        sc = O.scene_from_golden(name, own_bvh=own)
        st = O.Stats()
        hits = sc.trace(t["rays"], st)
        ref = t["hits"]
        assert np.array_equal(hits["hit"], ref["hit"])
        for f in O.HIT_DT.names:
            assert hits[f].tobytes() == ref[f].tobytes(), (name, f)
        assert st.real_rays == len(ref) and st.nodes >= len(ref)


def test_trace_probe_ray():
    # SURVEY.md 8(c) probe
    sc = O.scene_from_golden("cornell_obj")
    h = sc.trace(np.array([[0, 5, 20, 0, 0, -1.0]]))[0]
    assert h["hit"] == 1 and h["faceID"] == 7 and h["materialID"] == 0
    assert (h["t"], h["u"], h["v"]) == (24.591497079797261, 0.47423849825589276, 0.016982553970630036)
    assert tuple(h["normal"]) == (0.0, 0.023403813562830805, -0.99972609324290229)


RENDERS = ["render_cornell_obj_64_plane_2pass", "render_cornell_obj_64_noplane", "render_cornell_obj_128x96_plane",
           "render_cornell_eson_48_plane", "render_cornell_obj_40x56_view2", "render_teapot_obj_64x48_plane"]


@pytest.mark.parametrize("name", RENDERS)
def test_render_reference_stream_bit_exact(name):
    r = O.load_golden(name)
    mesh = "cornell_eson" if "eson" in name else ("teapot_obj" if "teapot" in name else "cornell_obj")
    sc = O.scene_from_golden(mesh)
    W, H = int(r["W"]), int(r["H"])
    frame = O.camera_frame(r["eye"], r["lookat"], r["up"], r["quat"], 45.0, W, H)
    plane = sc.plane() if int(r["plane"]) else None
    state = np.array(O.REFERENCE_SEED, "<u4")
    count = np.zeros((H, W), "<i4")
    for p in range(int(r["passes"])):
        img, count, st, _ = sc.render(frame, W, H, 16, 1, plane, O.RNG_STREAM, stream_state=state, count=count)
        assert img.tobytes() == r["images"][p].tobytes(), "pass %d differs" % p
        assert st["garbage_hits"] == 0
        # a path makes 1 or maxPathLength Trace() calls
        assert st["paths"] == W * H and (st["trace_calls"] - st["paths"]) % 15 == 0
    assert np.array_equal(count, r["count"])


@pytest.mark.parametrize("name", ["render_cornell_obj_64_plane_step2_2pass", "render_cornell_obj_60x48_noplane_step4"])
def test_render_step_reference_stream_bit_exact(name):
    """Render(step > 1) (render.cc:657-696): one path per block, block fill, count += 3 per pixel and call -- the oracle in
    the reference's serial stream against the reference's own images, call after call."""
    r = O.load_golden(name)
    osc = O.scene_from_golden("cornell_obj")
    W, H, passes, step = int(r["W"]), int(r["H"]), int(r["passes"]), int(r["step"])
    frame = O.camera_frame(r["eye"], r["lookat"], r["up"], r["quat"], 45.0, W, H)
    plane = osc.plane() if int(r["plane"]) else None
    state = np.array(O.REFERENCE_SEED, "<u4")
    count = np.zeros((H, W), "<i4")
    for p in range(passes):
        img, count, _, _ = osc.render_step(frame, W, H, step, 16, plane, O.RNG_STREAM, stream_state=state, count=count)
        assert img.tobytes() == r["images"][p].tobytes(), (name, p)
    assert np.array_equal(count, r["count"]) and int(count.min()) == 3 * passes


@pytest.mark.parametrize("name", ["aov_cornell_normal_64x48", "aov_teapot_normal_72x40", "aov_teapot_uv_64x48",
                                  "aov_cornell_uv_32x24"])
def test_show_normal_and_show_uv_reference_stream_bit_exact(name):
    """ShowNormal / ShowUV (render.cc:458-516): the oracle in the reference's serial stream against images produced by the
    reference's own functions (oracle/ref_aov_driver.cc #includes the unmodified render.cc)."""
    r = O.load_golden(name)
    osc = O.scene_from_golden("teapot_obj" if "teapot" in name else "cornell_obj")
    W, H = int(r["W"]), int(r["H"])
    frame = O.camera_frame(r["eye"], r["lookat"], width=W, height=H)
    img, st, _ = osc.render_aov(frame, W, H, int(r["mode"]), O.RNG_STREAM, stream_state=np.array(O.REFERENCE_SEED, "<u4"))
    assert img.tobytes() == r["image"].tobytes() and st["real_rays"] == W * H
    if "teapot" in name:
        assert (img != 0).any()


def test_render_512_digest():
    r = O.load_golden("render_cornell_obj_512_plane_digest")
    sc = O.scene_from_golden("cornell_obj")
    frame = O.camera_frame(r["eye"], r["lookat"], width=512, height=512)
    state = np.array(O.REFERENCE_SEED, "<u4")
    img, _, st, _ = sc.render(frame, 512, 512, 16, 1, sc.plane(), O.RNG_STREAM, stream_state=state)
    assert hashlib.sha256(img.tobytes()).digest() == r["sha256"].tobytes()
    assert np.array_equal(img[::64], r["rows"])
    # SURVEY.md section 0/6 scalars of the reference default config
    assert int((img[..., 0] != 0).sum()) == 219558
    assert st["trace_calls"] == 3706279 and st["real_rays"] == 1035072
    assert st["garbage_nodes"] == st["trace_calls"] - st["real_rays"] == 2671207
    assert st["garbage_hits"] == 0


PANOS = ["pano_cornell_stereo_96x64", "pano_cornell_mono_80x40", "pano_cornell_stereo_50x37_view2", "pano_teapot_mono_64x32"]


def pano_setup(name):
    r = O.load_golden(name)
    sc = O.scene_from_golden("teapot_obj" if "teapot" in name else "cornell_obj")
    W, H = int(r["W"]), int(r["H"])
    origin = O.camera_frame(r["eye"], r["lookat"], r["up"], r["quat"], 45.0, W, H)[:3]
    return r, sc, W, H, origin


@pytest.mark.parametrize("name", PANOS)
def test_panoramic_reference_stream_bit_exact(name):
    """mo_render_panoramic (RenderPanoramic + PathTraceEnv + Generate[Stereo]EnvRay) against the reference's own images."""
    r, sc, W, H, origin = pano_setup(name)
    state = np.array(O.REFERENCE_SEED, "<u4")
    img, count, st, states = sc.render_panoramic(origin, W, H, int(r["stereo"]), 16, 10, O.RNG_STREAM, stream_state=state,
                                                 want_states=True)
    assert img.tobytes() == r["image"].tobytes()
    assert np.array_equal(count, r["count"]) and int(count[0, 0]) == 10
    assert st["garbage_hits"] == 0
    assert st["paths"] == 10 * W * H and (st["trace_calls"] - st["paths"]) % 15 == 0  # 1 or 16 Trace() calls per path
    # per-pixel start states replayed in parallel give the same image (the bridge to the GPU)
    img2, _, st2, _ = sc.render_panoramic(origin, W, H, int(r["stereo"]), 16, 10, O.RNG_TABLE, rng_states=states, nthreads=4)
    assert img2.tobytes() == img.tobytes() and st2["real_rays"] == st["real_rays"]


def test_env_rays_cover_the_sphere():
    """GenerateEnvRay: unit directions, v = 0 is +Y, u wraps around Y; stereo eyes sit 0.5 off the origin, left eye in the
    top half of the frame, toed in (camera.cc:242-329)."""
    o = np.array([1.0, 2.0, 3.0])
    r = O.generate_env_ray(o, 64, 32, 0, 0.0, 0.0)
    assert np.allclose(r[:3], o) and np.allclose(r[3:], (0, 1, 0))
    r = O.generate_env_ray(o, 64, 32, 0, 16.0, 16.0)  # phi = pi/2, theta = pi/2
    assert np.allclose(r[3:], (0, 0, 1), atol=1e-15)
    top = O.generate_env_ray(o, 64, 32, 1, 0.0, 8.0)      # left eye, theta = pi/2, phi = 0: looks along +X
    bot = O.generate_env_ray(o, 64, 32, 1, 0.0, 24.0)     # right eye, same direction
    assert np.isclose(np.linalg.norm(top[:3] - o), 0.5) and np.isclose(np.linalg.norm(bot[:3] - o), 0.5)
    assert np.allclose(top[:3] - o, -(bot[:3] - o))
    assert np.isclose(np.linalg.norm(top[3:]), 1.0) and top[5] * bot[5] < 0  # toed in: opposite z components


def test_table_mode_replays_stream():
    """Start states captured from a reference-stream run, replayed per pixel, give the same image: the bridge that
    lets a parallel (GPU) renderer be compared with the serial reference stream."""
    sc = O.scene_from_golden("cornell_obj")
    W = H = 64
    frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    state = np.array(O.REFERENCE_SEED, "<u4")
    img, _, _, states = sc.render(frame, W, H, 16, 2, sc.plane(), O.RNG_STREAM, stream_state=state, want_states=True)
    img2, _, _, _ = sc.render(frame, W, H, 16, 2, sc.plane(), O.RNG_TABLE, rng_states=states, nthreads=4)
    assert img.tobytes() == img2.tobytes()
    g = O.load_golden("render_cornell_obj_64_plane_2pass")
    assert np.array_equal(img, g["images"][0] + g["images"][1])
