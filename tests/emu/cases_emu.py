"""TEST INFRASTRUCTURE: cases that run the library's kernels on the wave64 emulator (tests/emu) against the oracle.  Run by
tests/test_emu_cpu.py in a child interpreter with MALLIE_MGPU_LIB = tests/emu/libmallie_mgpu_emu.so, MALLIE_ALLOW_EMULATOR=1.
The whole `-m gpu` suite can be pointed at the emulator the same way (profiles/r5_emulator_suite.txt says which tests pass there)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import mallie_amd as M  # noqa: E402
import oracle_lib as O  # noqa: E402
from mallie_amd.scenes import suzanne_grid  # noqa: E402

FIELDS = ("real_rays", "nodes", "tris", "trace_calls")


def test_this_is_the_emulator():
    assert "libmallie_mgpu_emu" in os.path.basename(M.lib_path()) and M.device_count() >= 1


def _golden_scene(name):
    g = O.load_golden(name)
    return M.Scene(g["verts"], g["faces"], g["matIDs"], g["normals"] if g["has_normals"] else None, O.golden_uvs(g), g["nodes"], g["indices"])


@pytest.mark.parametrize("kernel", ["sm_lds", "sm", "v1"])
def test_render_kernels_vs_oracle(kernel, monkeypatch):
    """k_render_sm with the scene in LDS (leaf hints, staged primary rays), the same walk through the wide records in HBM, and the
    first ray-synchronous kernel: frames byte-equal to the oracle's, counters equal (primary rays only: exact)."""
    monkeypatch.setenv("MGPU_RENDER_KERNEL", kernel)
    sc, osc = _golden_scene("cornell_obj"), O.scene_from_golden("cornell_obj")
    W, H = 72, 48
    frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    for (mpl, passes, win) in [(5, 3, None), (1, 2, None), (4, 1, (5, 3, 61, 40))]:
        img, cnt, st = sc.render(frame, W, H, mpl, passes, osc.plane(), M.RNG_HASH, seed=42, pass_base=3, window=win)
        oimg, ocnt, ost, _ = osc.render(frame, W, H, mpl, passes, osc.plane(), O.RNG_HASH, seed=42, pass_base=3, window=win)
        assert img.tobytes() == oimg.tobytes() and np.array_equal(cnt, ocnt), (kernel, mpl, passes, win)
        if mpl == 1:
            assert all(st[f] == ost[f] for f in FIELDS), (st, ost)
        else:
            assert st["real_rays"] == ost["real_rays"] and st["trace_calls"] == ost["trace_calls"]
            assert abs(st["nodes"] - ost["nodes"]) <= 2e-3 * ost["nodes"] and abs(st["tris"] - ost["tris"]) <= 2e-3 * ost["tris"]


@pytest.mark.parametrize("block", ["640", "320"])
def test_hand_partitioned_five_wave_kernel(block, monkeypatch):
    """k_render_w5 (mgpu_render_w5.hip): frames and counters word for word those of k_render_sm's HBM-resident walk, and the
    oracle's frame."""
    c = O.load_golden("cornell_obj")
    verts, faces, mats, normals = suzanne_grid(c["verts"], c["faces"], 3)
    W, H = 120, 72
    frame = M.camera_frame((0.0, 40.0, 80.0), (0.0, 0.0, 0.0), width=W, height=H)
    ref = M.Scene(verts, faces, mats, normals, None)
    monkeypatch.setenv("MGPU_W5", "1")
    monkeypatch.setenv("MGPU_W5_BLOCK", block)
    sc = M.Scene(verts, faces, mats, normals, None)
    plane = ref.plane()
    for (mpl, passes, win) in [(5, 3, None), (9, 1, None), (3, 2, (3, 5, 111, 67))]:
        monkeypatch.setenv("MGPU_W5", "1")
        img, cnt, st = sc.render(frame, W, H, mpl, passes, plane, M.RNG_HASH, seed=11, pass_base=2, window=win)
        monkeypatch.setenv("MGPU_W5", "0")
        rimg, rcnt, rst = ref.render(frame, W, H, mpl, passes, plane, M.RNG_HASH, seed=11, pass_base=2, window=win)
        assert img.tobytes() == rimg.tobytes() and np.array_equal(cnt, rcnt), (mpl, passes, win)
        assert all(st[f] == rst[f] for f in FIELDS + ("paths",)), (st, rst)
    osc = O.OracleScene(verts, faces, mats, normals, None)
    monkeypatch.setenv("MGPU_W5", "1")
    img, _, _ = sc.render(frame, W, H, 4, 2, plane, M.RNG_HASH, seed=5)
    oimg = osc.render(frame, W, H, 4, 2, osc.plane(), O.RNG_HASH, seed=5)[0]
    assert img.tobytes() == oimg.tobytes()


def test_reference_stream_resolution_with_the_analytic_horizon(monkeypatch):
    """MGPU_RNG_STREAM through the chip-wide resolution (classification incl. the plane's horizon band, rounds, verification)
    against the one-workgroup walk: tables, stream state and frames word for word; the oracle's frame in the reference's stream."""
    sc, osc = _golden_scene("cornell_obj"), O.scene_from_golden("cornell_obj")
    W, H, mpl = 96, 64, 5
    frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    plane = sc.plane()
    monkeypatch.setenv("MGPU_STREAM_SERIAL", "1")
    img0, _, st0, state0, states0 = sc.render_stream(frame, W, H, mpl, 2, plane, want_states=True)
    monkeypatch.delenv("MGPU_STREAM_SERIAL")
    img1, _, st1, state1, states1 = sc.render_stream(frame, W, H, mpl, 2, plane, want_states=True)
    assert np.array_equal(states1, states0) and np.array_equal(state1, state0) and img1.tobytes() == img0.tobytes()
    ss = sc.stream_stats()
    assert ss["uncertain_pixels"] >= W  # the horizon row is uncertain from the first call
    oimg = osc.render(frame, W, H, mpl, 2, osc.plane(), O.RNG_STREAM, stream_state=np.array(O.REFERENCE_SEED, "<u4"))[0]
    assert img1.tobytes() == oimg.tobytes()


@pytest.mark.parametrize("threads", ["0", "1"])
def test_several_ranks_in_one_process(threads, monkeypatch):
    """mgpu_frame_* with four ranks sharing the (emulated) device through the copy transport, the launch phase enqueued by the
    caller's thread or by one thread per member (EnqueuePool): every assembled frame equals the single-launch frame."""
    monkeypatch.setenv("MGPU_FRAME_TRANSPORT", "copy")
    monkeypatch.setenv("MGPU_FRAME_ENQUEUE_THREADS", threads)
    world = 4
    scenes = [_golden_scene("cornell_obj") for _ in range(world)]
    W, H, mpl, passes = 80, 53, 4, 2
    cam = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    plane = scenes[0].plane()
    fr = M.Frame(scenes, [0] * world, W, H, strip_h=8, frames_in_flight=3)
    slots = [fr.render(cam, mpl, passes, plane, seed=9, pass_base=k * passes) for k in range(2)]
    frames = [fr.wait(s, to_host=True) for s in slots]
    slots = fr.render_batch(cam, mpl, passes, 3, plane, seed=9, pass_base=2 * passes)
    frames += [fr.wait(s, to_host=True) for s in slots]
    for k, img in enumerate(frames):
        ref = scenes[0].render(cam, W, H, mpl, passes, plane, M.RNG_HASH, seed=9, pass_base=k * passes)[0]
        assert img.tobytes() == ref.tobytes(), (threads, k)
    fr.close()


@pytest.mark.parametrize("kernel", ["v1", "sm"])
def test_batched_trace_vs_reference_records(kernel, monkeypatch):
    monkeypatch.setenv("MGPU_TRACE_KERNEL", kernel)
    sc = _golden_scene("cornell_obj")
    t = O.load_golden("trace_cornell_obj")
    out, hit = sc.trace(t["rays"][:600])
    assert np.array_equal(hit, t["hits"]["hit"][:600].astype("u1"))
    for name in ("t", "u", "v", "faceID"):
        m = hit.astype(bool)
        assert np.array_equal(out[name][m], t["hits"][name][:600][m]), name


def test_one_ray_calls_through_the_resident_server():
    """k_trace_server -- sixteen resident single-wave workgroups polling a mailbox in mapped host memory -- runs on the emulator as a
    RESIDENT kernel: all workgroups alive, on a thread of its own, while the callers' threads post rays and spin on the
    acknowledgements (mgpu_api.hip, trace_served).  Records of one-ray calls from one and from four threads equal the batched
    kernel's and the reference's goldens."""
    import threading
    sc = _golden_scene("cornell_obj")
    t = O.load_golden("trace_cornell_obj")
    rays = t["rays"][:160]
    ref, ref_hit = sc.trace(rays)
    out, hit = sc.trace_calls(rays[:40], per_call=1)
    assert np.array_equal(hit, ref_hit[:40]) and out.tobytes() == ref[:40].tobytes()
    st = sc.trace_server_stats()
    assert st["calls"] == 40 and st["launches"] >= 1
    res = [None] * 4

    def work(k):
        res[k] = sc.trace_calls(rays[k::4], per_call=1)
    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for k in range(4):
        assert np.array_equal(res[k][1], ref_hit[k::4]) and res[k][0].tobytes() == ref[k::4].tobytes(), k
    m = ref_hit.astype(bool)
    assert np.array_equal(ref["t"][m], t["hits"]["t"][:160][m]) and np.array_equal(ref["faceID"][m], t["hits"]["faceID"][:160][m])
    # a render call retires the live launch first
    W, H = 32, 24
    frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    img = sc.render(frame, W, H, 3, 1, sc.plane(), M.RNG_HASH, seed=3)[0]
    assert not sc.trace_server_stats()["alive"]
    oimg = O.scene_from_golden("cornell_obj").render(frame, W, H, 3, 1, sc.plane(), O.RNG_HASH, seed=3)[0]
    assert img.tobytes() == oimg.tobytes()
    sc.close()


def test_five_wave_kernel_on_a_deep_tree(monkeypatch):
    """k_render_w5 keeps five far-child entries per lane in LDS; deeper ones go to the lane's column in HBM.  A chain of shells of
    geometrically growing size seen from inside gives a tree of depth 256 (minLeafPrimitives 1) whose walks hold far more: frames and
    counters equal k_render_sm's (eight entries in LDS) and the oracle's."""
    monkeypatch.setenv("MGPU_RENDER_KERNEL", "sm")
    rng = np.random.default_rng(7)
    V, F = [], []
    for k in range(70):
        s = 1.25 ** k
        c = np.array([s, 0.15 * s * rng.normal(), 0.15 * s * rng.normal()])
        for _ in range(2):
            a = c + 0.25 * s * np.array([0.0, rng.normal(), rng.normal()])
            V += [a, a + 0.2 * s * np.array([0.02, 1.0, 0.1]), a + 0.2 * s * np.array([-0.02, 0.1, 1.0])]
            F.append([len(V) - 3, len(V) - 2, len(V) - 1])
    verts, faces = np.array(V), np.array(F, "u4")
    nodes, idx, st = M.bvh_build(verts, faces, minLeaf=1)
    assert st["maxTreeDepth"] > 40
    ref = M.Scene(verts, faces, None, None, None, nodes, idx)
    W, H = 96, 64
    frame = M.camera_frame((-2.0, 0.0, 0.0), (10.0, 0.0, 0.0), width=W, height=H)
    rimg, rcnt, rst = ref.render(frame, W, H, 6, 2, None, M.RNG_HASH, seed=3)
    monkeypatch.setenv("MGPU_W5", "1")
    sc = M.Scene(verts, faces, None, None, None, nodes, idx)
    img, cnt, s5 = sc.render(frame, W, H, 6, 2, None, M.RNG_HASH, seed=3)
    assert img.tobytes() == rimg.tobytes() and all(s5[f] == rst[f] for f in FIELDS + ("paths",)), (s5, rst)
    assert s5["nodes"] > 5 * s5["real_rays"]
    osc = O.OracleScene(verts, faces, None, None, None, nodes, idx)
    oimg, _, ost, _ = osc.render(frame, W, H, 6, 2, None, O.RNG_HASH, seed=3)
    assert img.tobytes() == oimg.tobytes() and ost["nodes"] == s5["nodes"] and ost["tris"] == s5["tris"]


def test_five_wave_kernel_shares_a_render_slot_with_the_default_kernel(monkeypatch):
    """k_render_sm and k_render_w5 share a scene's render slot and its overflow columns with DIFFERENT column pitches (depth - 8 and
    depth - 5 entries per lane).  Round 5 remembered the buffer's capacity in lanes: a k_render_sm launch (here: a path length above
    k_render_w5's 255 makes the call fall back) sized it, and a later k_render_w5 launch with fewer lanes but the wider pitch wrote past
    it -- for trees 9 to 12 deep only (advisor, round 5).  A line of 4 096 triangles seen end-on (tree depth 11, every box on the way is
    hit, the stacks fill) does it: under the emulator's ASan build the round-5 library reports a heap-buffer-overflow in
    WStackP<5>::put here (profiles/r6_w5_slot_sharing_asan.txt), this one does not; in this plain run the frames must be k_render_sm's."""
    monkeypatch.setenv("MGPU_RENDER_KERNEL", "sm")
    rng = np.random.default_rng(1)
    V, F = [], []
    for k in range(4096):
        z, j = -0.5 * k, rng.normal(size=3) * 0.05
        V += [[-0.4 + j[1], -0.4, z + j[0]], [0.4, -0.3 + j[2], z], [-0.3, 0.4, z]]
        F.append([3 * k, 3 * k + 1, 3 * k + 2])
    verts, faces = np.array(V, np.float64), np.array(F, "u4")
    nodes, idx, st = M.bvh_build(verts, faces, minLeaf=4)
    assert 9 <= st["maxTreeDepth"] <= 12, st
    W, H = 80, 8
    frame = M.camera_frame((0.0, 0.0, 5.0), (0.0, 0.0, -10.0), width=W, height=H, fov=1.0)
    monkeypatch.setenv("MGPU_W5", "0")
    ref = M.Scene(verts, faces, None, None, None, nodes, idx)
    rlong = ref.render(frame, W, H, 300, 1, None, M.RNG_HASH, seed=5)
    rimg, rcnt, rst = ref.render(frame, W, H, 4, 4, None, M.RNG_HASH, seed=3)
    assert rst["nodes"] > 15 * rst["real_rays"]  # the walks are deep
    monkeypatch.setenv("MGPU_W5", "1")
    sc = M.Scene(verts, faces, None, None, None, nodes, idx)
    long_img, _, long_st = sc.render(frame, W, H, 300, 1, None, M.RNG_HASH, seed=5)  # falls back to k_render_sm: sizes the slot's columns
    img, cnt, s5 = sc.render(frame, W, H, 4, 4, None, M.RNG_HASH, seed=3)            # k_render_w5: fewer lanes, wider pitch
    assert long_img.tobytes() == rlong[0].tobytes() and long_st["nodes"] == rlong[2]["nodes"]
    assert img.tobytes() == rimg.tobytes() and all(s5[f] == rst[f] for f in FIELDS + ("paths",)), (s5, rst)


def test_kernels_ran_from_their_gfx950_machine_code():
    """With MGPU_EMU_ISA=<hipcc -S dumps> (tests/emu/isa_interp.cc) the launches of the cases above executed hipcc's gfx950 INSTRUCTION STREAMS,
    not the host-compiled C++: the render kernel again, counted -- its frame is the oracle's and the interpreter's instruction counters moved by
    what a frame of that size costs (30 .. 80 VALU wave-instructions per ray at this size; 35 on a full frame, like the hardware's counter)."""
    import ctypes
    if not os.environ.get("MGPU_EMU_ISA"):
        pytest.skip("MGPU_EMU_ISA is not set: the kernels ran as host-compiled C++")
    L = ctypes.CDLL(os.environ["MALLIE_MGPU_LIB"])
    cnt = (ctypes.c_ulonglong * 16).in_dll(L, "isa_counters")
    g = O.load_golden("cornell_obj")
    sc = M.Scene(g["verts"], g["faces"], g["matIDs"], g["normals"], None)
    osc = O.scene_from_golden("cornell_obj")
    W, H = 96, 64
    frame = M.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    before = [int(x) for x in cnt]
    img, _, st = sc.render(frame, W, H, 5, 3, sc.plane(), M.RNG_HASH, seed=7)
    after = [int(x) for x in cnt]
    oimg, _, ost, _ = osc.render(frame, W, H, 5, 3, osc.plane(), O.RNG_HASH, seed=7)
    assert img.tobytes() == oimg.tobytes() and (st["nodes"], st["tris"]) == (ost["nodes"], ost["tris"])
    valu, salu, launches = after[0] - before[0], after[1] - before[1], after[8] - before[8]
    assert launches >= 2, "the render launch did not go through the ISA interpreter"
    assert 30.0 < valu / st["real_rays"] < 80.0 and 0.4 < salu / valu < 0.7, (valu, salu, st["real_rays"])
    sc.close()
