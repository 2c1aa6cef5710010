"""CPU-only checks of the product's host side (mallie_amd + libmallie_mgpu.so): the C-ABI library loads and exports
every symbol include/mgpu.h declares, the host-side pieces of the path (camera frame, BVH build, plane, RNG seeding)
agree bit-for-bit with the reference goldens and with the oracle, and compute calls fail loudly without a GPU."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

import mallie_amd as M
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_strip_ownership_matches_the_python_partition():
    """mgpu_frame_rows (the C ABI's multi-GPU frame) and mallie_amd.frame.strip_rows (the torch.distributed path) cut a
    frame the same way, for ragged heights and strip heights too."""
    from mallie_amd.frame import strip_rows
    for H in (1, 7, 8, 9, 135, 1080, 2160, 1081):
        for sh in (1, 5, 8, 13):
            for world in (1, 2, 3, 4, 8):
                rows = [M.frame_rows(H, sh, world, r) for r in range(world)]
                assert rows == [len(strip_rows(H, world, r, sh)) for r in range(world)] and sum(rows) == H
    assert M.frame_rows(10, 8, 2, 2) == -1 and M.frame_rows(10, 0, 2, 0) == -1


def test_frame_exchange_plan_reassembles_any_frame():
    """The exchange plan of the C ABI's multi-GPU frame (mgpu_frame_plan; both the ncclSend and the ncclRecv loops walk it):
    executed here with numpy copies for world sizes 2..8 on ragged frames, it moves every rank's local strips to exactly the
    rows the torch.distributed path (mallie_amd/frame.py) assigns to that rank, and tiles the frame without gaps or overlap."""
    from mallie_amd.frame import strip_rows
    rng = np.random.default_rng(3)
    for W, H, sh in ((7, 61, 8), (16, 64, 8), (5, 203, 13), (3, 9, 1), (4, 1080, 8)):
        frame = rng.random((H, W, 3)).astype("<f4")
        for world in (2, 3, 4, 8):
            out = np.full(H * W * 3, np.nan, "<f4")
            covered = np.zeros(H * W * 3, bool)
            for r in range(world):
                rows = strip_rows(H, world, r, sh)
                local = frame[rows].reshape(-1)                      # what rank r's kernel leaves in its strip buffer
                lo, fo, cnt = M.frame_plan(W, H, sh, world, r)
                assert int(cnt.sum()) == local.size == 3 * W * M.frame_rows(H, sh, world, r)
                for a, b, c in zip(lo, fo, cnt):
                    a, b, c = int(a), int(b), int(c)
                    assert not covered[b:b + c].any()
                    covered[b:b + c] = True
                    out[b:b + c] = local[a:a + c]                    # ncclSend(local + a, c) -> ncclRecv(frame + b, c)
            assert covered.all() and out.tobytes() == frame.tobytes(), (W, H, sh, world)


def test_block_exchange_plan_reassembles_any_frame():
    """The default exchange (MGPU_EXCHANGE_BLOCK): every rank's strip buffer is ONE message into rank 0's staging area and one
    strided 2-D copy (+ a plain copy for a partial last strip) deals it to the frame.  mgpu_frame_block_plan hands out the very
    numbers the device code passes to ncclRecv / hipMemcpy2DAsync / hipMemcpyAsync; executed here with numpy on byte arrays for
    world sizes 1 (exchange forced) .. 8 on ragged frames, they must rebuild the frame exactly, staging areas must tile without
    overlap, and rank 0's own strips (placed from its strip buffer, no message) must land where its plan says."""
    from mallie_amd.frame import strip_rows
    rng = np.random.default_rng(4)

    def copy2d(dst, dst_off, dst_pitch, src, src_off, src_pitch, width, height):
        for r in range(height):
            dst[dst_off + r * dst_pitch: dst_off + r * dst_pitch + width] = src[src_off + r * src_pitch: src_off + r * src_pitch + width]

    for W, H, sh in ((7, 61, 8), (16, 64, 8), (5, 203, 13), (3, 9, 1), (4, 1080, 8), (2, 5, 8)):
        frame = rng.random((H, W, 3)).astype("<f4")
        for world, force in ((1, True), (2, False), (3, False), (4, False), (8, False)):
            out = np.zeros(H * W * 12, "u1")
            written = np.zeros(H * W * 12, bool)
            n_stage = sum(3 * W * M.frame_rows(H, sh, world, r) for r in range(0 if force else 1, world))
            staging = np.zeros(n_stage * 4, "u1")
            staged = np.zeros(n_stage, bool)
            for r in range(world):
                rows = strip_rows(H, world, r, sh)
                local = np.ascontiguousarray(frame[rows]).view("u1").reshape(-1)   # rank r's strip buffer, bytes
                p = M.frame_block_plan(W, H, sh, world, r, force)
                assert p["msg_floats"] * 4 == local.size
                if r == 0 and not force:
                    src = local                                                    # own strips: straight from the strip buffer
                else:
                    a, n = p["staging_off"], p["msg_floats"]                       # ncclSend(local, n) -> ncclRecv(staging + a, n)
                    assert not staged[a:a + n].any()
                    staged[a:a + n] = True
                    staging[4 * a:4 * (a + n)] = local
                    src = staging[4 * a:4 * (a + n)]
                mark = np.ones(local.size, "u1")
                for buf, s_ in ((out, src), (written.view("u1"), mark)):
                    copy2d(buf, p["dst_off"], p["dst_pitch"], s_, 0, p["src_pitch"], p["width"], p["height"])
                    if p["tail_bytes"]:
                        buf[p["tail_dst_off"]:p["tail_dst_off"] + p["tail_bytes"]] = s_[p["tail_src_off"]:p["tail_src_off"] + p["tail_bytes"]]
            assert staged.all() and written.all(), (W, H, sh, world)
            assert out.tobytes() == frame.tobytes(), (W, H, sh, world)


def test_library_exports_every_declared_symbol():
    """include/mgpu.h is the drop-in boundary, include/mgpu_internal.h the instrumentation this repository's own tests and tools use:
    every function either declares is exported, and no diagnostic is declared in the boundary header."""
    lib = ctypes.CDLL(M.lib_path())
    for header, at_least in (("mgpu.h", 15), ("mgpu_internal.h", 5)):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        names = sorted(set(re.findall(r"\b(mgpu_[a-z_0-9]+)\s*\(", hdr)))
        assert len(names) >= at_least, names
        for n in names:
            assert hasattr(lib, n), "libmallie_mgpu.so does not export %s (%s)" % (n, header)
        if header == "mgpu.h":
            assert not [n for n in names if n.startswith("mgpu_debug_") or n in ("mgpu_trace_calls_measure", "mgpu_occupancy_read")], names
    assert M.abi_version() == 1


def test_pod_layouts_match_reference_sizes():
    # sizeof measured from the reference build
    assert M.NODE_DT.itemsize == 64 and M.RAY_DT.itemsize == 88 and M.ISECT_DT.itemsize == 184
    assert M.ISECT_DT.fields["position"][1] == 48 and M.ISECT_DT.fields["normal"][1] == 96
    assert M.ISECT_DT.fields["texcoord"][1] == 168 and M.RAY_DT.fields["dirSign"][1] == 72


def test_camera_frame_matches_reference_goldens():
    g = O.load_golden("camera")
    for cfg, frame in zip(g["cfg"], g["frames"]):
        f = M.camera_frame(cfg[3:6], cfg[6:9], cfg[9:12], cfg[12:16], cfg[2], int(cfg[0]), int(cfg[1]))
        assert f.tobytes() == frame.tobytes(), cfg
        assert f.tobytes() == O.camera_frame(cfg[3:6], cfg[6:9], cfg[9:12], cfg[12:16], cfg[2], int(cfg[0]),
                                             int(cfg[1])).tobytes()


@pytest.mark.parametrize("name", ["cornell_obj", "cornell_eson", "teapot_obj"])
def test_bvh_build_matches_reference_goldens(name):
    g = O.load_golden(name)
    nodes, idx, st = M.bvh_build(g["verts"], g["faces"])
    assert np.array_equal(idx, g["indices"])
    assert nodes.tobytes() == g["nodes"].tobytes()
    assert st["numLeafNodes"] + st["numBranchNodes"] == len(nodes)


def test_bvh_build_matches_oracle_on_synthetic_meshes():
    rng = np.random.default_rng(5)
    for nf in (1, 15, 16, 17, 200, 3000):
        verts = rng.normal(size=(3 * nf, 3)).round(3)
        verts[: nf // 2] *= 0.0  # degenerate cluster: exercises the median fallback of a failed partition
        faces = rng.integers(0, len(verts), (nf, 3)).astype("u4")
        a = M.bvh_build(verts, faces)
        b = O.bvh_build(verts, faces)
        assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and a[2] == b[2], nf
    # non-default options
    a = M.bvh_build(verts, faces, 0.35, 4, 6, 16)
    b = O.bvh_build(verts, faces, 0.35, 4, 6, 16)
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and a[2]["maxTreeDepth"] <= 6


def test_plane_and_seed_helpers_match_oracle():
    sc = O.scene_from_golden("cornell_obj")
    lo, hi = sc.bbox()
    assert M.plane_from_bbox(lo, hi).tobytes() == sc.plane().tobytes()
    for seed, p, px in [(1, 0, 0), (1, 15, 1920 * 1080 - 1), (0xDEADBEEFCAFE, 3, 77), (2 ** 64 - 1, 2 ** 32 - 1, 5)]:
        assert np.array_equal(M.hash_state(seed, p, px), O.hash_state(seed, p, px))


def test_compute_fails_loudly_without_gpu():
    if M.device_count() > 0:
        pytest.skip("a GPU is present")
    g = O.load_golden("cornell_obj")
    with pytest.raises(M.MgpuError) as e:
        M.Scene(g["verts"], g["faces"], g["matIDs"], g["normals"], None, g["nodes"], g["indices"])
    assert e.value.status == -2 and "no CPU path" in str(e.value)


def test_product_does_not_touch_the_oracle():
    """The shipped package must never import, link or call oracle/ (the checker)."""
    pkg = os.path.join(ROOT, "mallie_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cc", ".h")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "mallie_oracle" not in txt and "oracle_lib" not in txt and "libmallie_oracle" not in txt, fn
    out = os.popen("ldd %s" % M.lib_path()).read()
    assert "oracle" not in out


def test_tail_division_through_the_rounded_reciprocal_is_exact():
    """k_render_sm evaluates the post-miss tail's x / L (L <= 16) as q = x * y, q' = fma(fma(-q, L, x), y, q) with
    y = RN(1 / L) from the host (RenderParams::inv_len).  With a correctly rounded reciprocal that sequence yields the
    correctly rounded quotient (Markstein); checked here against IEEE division with exact rational arithmetic, on throughput-
    like values, random mantissas and mantissas next to the powers of two."""
    import math
    import random
    from fractions import Fraction

    def rn(fr):  # Fraction -> nearest double, ties to even
        if fr == 0:
            return 0.0
        sign, fr = (-1 if fr < 0 else 1), abs(fr)
        sh = 52 - (fr.numerator.bit_length() - fr.denominator.bit_length())
        m = fr * (Fraction(2) ** sh)
        if m < 2 ** 52:
            sh, m = sh + 1, m * 2
        if m >= 2 ** 53:
            sh, m = sh - 1, m / 2
        fl = m.numerator // m.denominator
        rem = m - fl
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (fl & 1)):
            fl += 1
        return sign * math.ldexp(fl, -sh)

    rng = random.Random(7)
    for L in range(1, 17):
        y = 1.0 / L
        assert y == rn(Fraction(1, L))
        for k in range(1500):
            if k % 3 == 0:
                a = rng.random() * 0.5 * (0.8 ** rng.randint(0, 16))
            elif k % 3 == 1:
                a = math.ldexp(rng.getrandbits(53) | (1 << 52), -53 - rng.randint(0, 40))
            else:
                m = (1 << 53) - 1 - rng.getrandbits(6) if rng.random() < 0.5 else (1 << 52) + rng.getrandbits(6)
                a = math.ldexp(m, -53 - rng.randint(0, 20))
            A = Fraction(a)
            q = rn(A * Fraction(y))
            r = A - Fraction(q) * L  # what fma(-q, L, a) returns: it is exactly representable
            assert Fraction(rn(r)) == r
            assert rn(Fraction(q) + r * Fraction(y)) == a / L, (a, L)


def test_tail_table_scales_with_a_power_of_two_throughput():
    """k_render_sm reads the post-miss tail of a path whose throughput is a power of two and whose multiplier is 0.5 (or absent) from
    RenderParams::tail_unit: throughput x (the tail loop's result for throughput 1).  The claim behind it -- scaling every operand of
    the loop by a power of two scales every rounded intermediate by it -- checked here by running the loop itself (Python floats are
    IEEE doubles; math.fma where the interpreter has it, exact rational arithmetic otherwise) for every (first length, maxPathLength,
    multiplier on / off) and throughputs 2^0 .. 2^-40 and 2^-900."""
    import math
    from fractions import Fraction

    if hasattr(math, "fma"):
        fma = math.fma
    else:
        def fma(a, b, c):  # correctly rounded a * b + c through exact rationals (float(Fraction) rounds to nearest even)
            return float(Fraction(a) * Fraction(b) + Fraction(c))

    def tail(thr, L0, mpl, mul):
        rad = 0.0
        L = L0
        while True:
            x, y, dl = thr * 0.5, 1.0 / L, float(L)
            q = x * y
            rad += fma(fma(-q, dl, x), y, q)
            if L >= mpl:
                break
            if mul:
                thr *= 0.5
            L += 1
        return rad

    for mpl in (2, 5, 9, 16):
        for L0 in range(1, mpl + 1):
            for mul in (False, True):
                unit = tail(1.0, L0, mpl, mul)
                for k in list(range(0, 41)) + [900]:
                    thr = math.ldexp(1.0, -k)
                    assert tail(thr, L0, mpl, mul) == thr * unit, (mpl, L0, mul, k)
    # and a throughput that is NOT a power of two does not scale like that in general (the kernel then runs the loop)
    assert any(tail(0.3 * math.ldexp(1.0, -k), 2, 16, True) != 0.3 * math.ldexp(1.0, -k) * tail(1.0, 2, 16, True) for k in range(8))


def test_bench_collects_its_own_counter_passes_when_the_committed_ones_are_of_another_build(tmp_path, monkeypatch):
    """bench.py's roofline must not depend on profiles/pmc_current.json carrying the loaded library's stamp: with a
    stamp of another build it runs rocprofv3 passes itself.  Here a stand-in `rocprofv3` on PATH writes the CSVs a real pass leaves
    behind (no GPU in this container); the entry bench.py builds from them must be per frame and carry the traced kernel average."""
    import importlib
    fake = tmp_path / "bin" / "rocprofv3"
    fake.parent.mkdir()
    fake.write_text('''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
d = a[a.index("-d") + 1]; os.makedirs(d, exist_ok=True)
frames = int(a[-1])
K = '"void mgpu::k_render_sm<unsigned char, true, 1024, true, true>(mgpu::DScene, mgpu::RenderParams)"'
if "--pmc" in a:
    ctrs = a[a.index("--pmc") + 1:a.index("--kernel-include-regex")]
    with open(os.path.join(d, "p_counter_collection.csv"), "w") as f:
        f.write('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\\n')
        for disp in range(frames):
            for c in ctrs:
                f.write('%d,%d,"Agent 2",1,1,1,262144,36,%s,1024,3072,8,64,0,112,"%s",%f,0,1\\n' % (disp, disp, K, c, 1000.0 + len(c)))
else:
    with open(os.path.join(d, "p_kernel_stats.csv"), "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\\n')
        f.write('%s,%d,%d,5000000.0,99.0,1,2,0.0\\n' % (K, frames, 5000000 * frames))
''')
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(fake.parent) + os.pathsep + os.environ["PATH"])
    monkeypatch.delenv("MALLIE_BENCH_SELF_PMC", raising=False)
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))  # no profiles/pmc_current.json there; the passes' CSVs land under tmp_path/gpurun_out
    (tmp_path / "tools").mkdir()
    import shutil
    shutil.copy(os.path.join(ROOT, "tools", "pmc_collect.py"), tmp_path / "tools" / "pmc_collect.py")
    monkeypatch.setitem(bench._SELF_PMC, "entries", {})
    monkeypatch.setitem(bench._SELF_PMC, "t0", None)
    monkeypatch.setitem(bench._SELF_PMC, "enabled", True)
    lib = os.path.join(ROOT, "mallie_amd", "libmallie_mgpu.so")
    e, why = bench.pmc_for("c2", lib)
    assert e is not None and "collected by THIS run" in why, why
    assert e["SQ_INSTS_VALU"] == 1000.0 + len("SQ_INSTS_VALU") and e["launches_per_frame"] == 1.0  # per frame, not per pass
    assert e["FETCH_SIZE"] == 1010.0 and e["WRITE_SIZE"] == 1010.0 and abs(e["traced_kernel_avg_ms"] - 5.0) < 1e-9
    assert bench.hbm_bytes(e) == int(2 * 1010 * 1024 + 1010 * 1024)
    e4, _ = bench.pmc_for("c5", lib)  # an HBM-resident extra: one frame per pass
    assert e4["FETCH_SIZE"] == 1010.0 and "traced_kernel_avg_ms" not in e4
    # switched off (how collect_pmc.sh runs bench.py under rocprofv3): nothing is spawned, the reason says so
    monkeypatch.setitem(bench._SELF_PMC, "entries", {})
    monkeypatch.setitem(bench._SELF_PMC, "enabled", False)
    e, why = bench.pmc_for("c2", lib)
    assert e is None and "switched off" in why


def test_enqueue_pool_hands_every_member_its_job_once_plain_and_under_tsan(tmp_path):
    """mallie_amd/csrc/mgpu_enqueue_pool.hpp (the multi-GPU frame object's launch-phase workers): tests/cpp/pool_driver.cc runs 3 spin
    windows x 3000 calls x 8 members -- parked and spinning workers, failing members -- built plain and with ThreadSanitizer."""
    import subprocess
    src = os.path.join(ROOT, "tests", "cpp", "pool_driver.cc")
    for name, flags in (("pool_plain", ["-O2"]), ("pool_tsan", ["-O1", "-g", "-fsanitize=thread"])):
        exe = str(tmp_path / name)
        subprocess.run(["g++", "-std=c++17", "-pthread"] + flags + [src, "-o", exe], check=True, capture_output=True)
        r = subprocess.run([exe, "stress"], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
        assert r.returncode == 0 and "pool stress ok" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[-3000:]
