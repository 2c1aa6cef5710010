"""Adversarial rays for the leaf hints -- TEST INFRASTRUCTURE (used by tests/test_hint_soundness_cpu.py and tests/test_gpu_parity.py).

TriangleIsect (bvh_accel.cc:595-638) guards its division only by the ABSOLUTE |det| >= 1024 eps.  A ray within ~1e-13 rad of a
triangle's plane therefore gets (u, v, t) that are off by up to ~|org - p0| |e1| |e2| / 100 in world units, and the reference
ACCEPTS rays whose true line passes well outside the triangle (the round-4 review constructed one).  A leaf hint that drops a
triangle because the ray misses a box around it must allow for exactly that (mallie_amd/csrc/mgpu_device.hpp, leaf_hint_make).

This module builds scenes that aim whole image rows of primary rays -- rays the render kernel generates itself from a table of RNG
start states -- into that band: a large right triangle T whose plane contains the eye up to a tilt of 1e-15 .. 1e-11 rad, T's box
corner p1 a few hundredths of T's size beside the rays, four small triangles far behind it (so that the five form one leaf whose
best split is "T | the rest").  The oracle (the pinned restatement of the reference) says which rays hit T.
"""
import numpy as np

import oracle_lib as O

EPS1024 = 2.220446049250313e-16 * 1024
DBL_MAX = 1.7976931348623157e+308


# ---- xorshift128 (render.cc:137-168): a start state whose first two outputs are given words -----------------------------
def state_for_draws(k1, k2):
    m = 0xFFFFFFFF
    w = (k1 ^ (k1 >> 19)) & m           # x = 0 -> t = 0 -> first output = w ^ (w >> 19), an involution on 32 bits
    c = (k2 ^ (k1 ^ (k1 >> 19))) & m     # second output = (k1 ^ (k1 >> 19)) ^ (t ^ (t >> 8)), t = y ^ (y << 11)
    t = (c ^ (c >> 8) ^ (c >> 16) ^ (c >> 24)) & m
    y = (t ^ (t << 11) ^ (t << 22)) & m
    return (0, y, 521288629, w)


def xorshift_outputs(state, n):
    x, y, z, w = [int(v) for v in state]
    m, out = 0xFFFFFFFF, []
    for _ in range(n):
        t = (x ^ (x << 11)) & m
        x, y, z = y, z, w
        w = ((w ^ (w >> 19)) ^ (t ^ (t >> 8))) & m
        out.append(w)
    return out


# ---- the reference's arithmetic, vectorised (IEEE double, the reference's operation order, no FMA) -----------------------
def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


def dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def triangle_isect(org, d, p0, e1, e2, tbest=DBL_MAX):
    """TriangleIsect for n rays against one triangle -> (accepted, t, u, v, det)."""
    p = cross(d, e2[None, :])
    det = dot(e1[None, :], p)
    with np.errstate(all="ignore"):
        inv = 1.0 / det
        s = org - p0[None, :]
        q = cross(s, e1[None, :])
        u = dot(s, p) * inv
        v = dot(q, d) * inv
        t = dot(e2[None, :], q) * inv
        ok = ~(np.abs(det) < EPS1024)
        ok &= ~((u < 0) | (u > 1)) & ~((v < 0) | (u + v > 1)) & ~((t < 0) | (t > tbest))
    return ok, t, u, v, det


def slab_plain(lo, hi, org, d, bt=DBL_MAX):
    """slab_t<true> of mgpu_device.hpp (the form a hint consultation uses) for n rays against one box."""
    inv = 1.0 / d
    l, h = (lo[None, :] - org) * inv, (hi[None, :] - org) * inv
    tmin = np.maximum(np.maximum(np.minimum(l[:, 0], h[:, 0]), np.minimum(l[:, 1], h[:, 1])), np.minimum(l[:, 2], h[:, 2]))
    tmax = np.minimum(np.minimum(np.maximum(l[:, 0], h[:, 0]), np.maximum(l[:, 1], h[:, 1])), np.maximum(l[:, 2], h[:, 2]))
    return (tmax > 0) & (tmin <= tmax) & (tmin <= bt)


def primary_rays(frame, W, words_u, words_v, rows):
    """PathTrace's eye rays (render.cc:387-391 + camera.cc:222-240) of pixels (x, rows[i]) for x < W: jitter words -> rays."""
    o, c, du, dv = frame[0:3], frame[3:6], frame[6:9], frame[9:12]
    ju = (words_u.astype(np.float64) * (1.0 / 4294967296.0) - 0.5).astype(np.float32)
    jv = (words_v.astype(np.float64) * (1.0 / 4294967296.0) - 0.5).astype(np.float32)
    u = (np.arange(W, dtype=np.float32)[None, :] + ju).astype(np.float64)
    v = (np.asarray(rows, dtype=np.float32)[:, None] + jv).astype(np.float64)
    d = np.stack([(c[k] + u * du[k] + v * dv[k]) - o[k] for k in range(3)], -1)
    ln = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])
    d = d * (1.0 / ln)[..., None]
    return np.broadcast_to(o, d.shape).copy(), d


# ---- the hint rule, restated for the test (mgpu_device.hpp: leaf_hint_make / leaf_hint_apply) -------------------------------
def _f32_down(x):
    f = x.astype(np.float32)
    return np.where(f.astype(np.float64) > x, np.nextafter(f, np.float32(-np.inf)), f).astype(np.float64)


def _f32_up(x):
    f = x.astype(np.float32)
    return np.where(f.astype(np.float64) < x, np.nextafter(f, np.float32(np.inf)), f).astype(np.float64)


def hint_boxes(tris, worth=0.85, rule="r5", centre=None, q=None, cone_max=0.2):
    """tris: (n, 3, 3) = p0, e1, e2 of a leaf's run.  -> (m, (loA, hiA, coneA), (loB, hiB, coneB)) or None; cone = (nbar[3], thr) as
    float32, or None under rule "r4".
    rule "r4": the round-4 library: pad = 2^-8 of the largest extent + 2^-20 of the largest coordinate, no second clause;
    rule "r5": mgpu_device.hpp (leaf_hint_make): a half whose normals fit a cone of threshold <= cone_max keeps the small pad
    2^-8 ext + 2^-40 (reach + largest coordinate) and the cone clause; any other half gets 9 / 1024 E_half reach and no cone.
    centre, q: the consulting rays' origins lie within q of centre (reach_of)."""
    n = len(tris)
    corners = np.stack([tris[:, 0], tris[:, 0] + tris[:, 1], tris[:, 0] + tris[:, 2]], 1)  # (n, 3, 3)

    def box(a, b):
        pts = corners[a:b].reshape(-1, 3)
        return pts.min(0), pts.max(0)

    def half_area(lo, hi):
        d = hi - lo
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0]

    whole = half_area(*box(0, n)) * n
    best, best_m = np.inf, 0
    for m in range(1, n):
        c = half_area(*box(0, m)) * m + half_area(*box(m, n)) * (n - m)
        if c < best:
            best, best_m = c, m
    if best_m == 0 or not best < worth * whole:
        return None
    out = [best_m]
    u, up = 2.0 ** -53, 1.0 + 2.0 ** -9
    for a, b in ((0, best_m), (best_m, n)):
        lo, hi = box(a, b)
        big = max(np.abs(lo).max(), np.abs(hi).max())
        ext = (hi - lo).max()
        if rule == "r4":
            pad = ext * 2.0 ** -8 + big * 2.0 ** -20
            out.append((_f32_down(lo - pad), _f32_up(hi + pad), None))
            continue
        reach = (q + np.sqrt((np.maximum(np.abs(lo - centre), np.abs(hi - centre)) ** 2).sum())) * up
        pad_geo = ext * 2.0 ** -8
        slack = 2.0 ** -20 * (np.abs(centre).max() + 2.0 * reach + big)
        nk = cross(tris[a:b, 1], tris[a:b, 2])
        ln = np.sqrt((nk ** 2).sum(1))
        ek = np.sqrt((tris[a:b, 1] ** 2).sum(1)) * np.sqrt((tris[a:b, 2] ** 2).sum(1)) * up
        live = ~(ln * up + 6.0 * u * ek < 2048.0 * u)
        ssum = np.zeros(3)
        for k in np.where(live & (ln > 0))[0]:
            ssum += (-1.0 if nk[k] @ ssum < 0 else 1.0) * nk[k] / ln[k]
        sl = np.sqrt((ssum ** 2).sum())
        nb = (ssum / sl).astype(np.float32) if sl > 0 else np.zeros(3, np.float32)
        thr = 0.0
        with np.errstate(all="ignore"):
            d_safe = 17.93 * u * reach * (ek[live].max() if live.any() else 0.0) / pad_geo
            for k in np.where(live)[0]:
                x = nk[k] / ln[k]
                rho = np.sqrt(min(((x - nb) ** 2).sum(), ((x + nb) ** 2).sum()))
                t = rho * up + (d_safe + 4.0 * u * ek[k]) / ln[k] * up + 2.0 ** -21
                thr = np.inf if not t == t else max(thr, t)
        thr32 = np.float32(thr)
        if float(thr32) < thr:
            thr32 = np.nextafter(thr32, np.float32(np.inf))
        pad = pad_geo + slack
        if not thr <= cone_max:
            pad = (9.0 / 1024.0) * (ek[live].max() if live.any() else 0.0) * reach + slack
            thr32 = np.float32(0.0)
        if not pad < np.inf:
            return None
        out.append((_f32_down(lo - pad), _f32_up(hi + pad), (nb, thr32)))
    return tuple(out)


def hint_keeps(half, org, d, bt=DBL_MAX):
    """leaf_hint_apply for one half (lo, hi, cone): True where the half stays in the run."""
    lo, hi, cone = half
    if cone is None:  # the round-4 library tested in double
        return slab_plain(lo, hi, org, d, bt)
    with np.errstate(all="ignore"):  # the float form of leaf_hint_apply: origin, 1 / d (a double division first) rounded to nearest
        of, jf = org.astype(np.float32), (1.0 / d).astype(np.float32)
        lof, hif = lo.astype(np.float32), hi.astype(np.float32)
        btf = np.float32(bt)
        if float(btf) < bt:
            btf = np.nextafter(btf, np.float32(np.inf))
        a, b = (lof[None, :] - of) * jf, (hif[None, :] - of) * jf
        tmin = np.maximum(np.maximum(np.minimum(a[:, 0], b[:, 0]), np.minimum(a[:, 1], b[:, 1])), np.minimum(a[:, 2], b[:, 2]))
        tmax = np.minimum(np.minimum(np.maximum(a[:, 0], b[:, 0]), np.maximum(a[:, 1], b[:, 1])), np.maximum(a[:, 2], b[:, 2]))
        keep = (tmax > 0) & (tmin <= tmax) & (tmin <= btf)
    if cone is not None:
        nb, thr = cone
        df = d.astype(np.float32)
        c = np.abs(df[:, 0] * nb[0] + (df[:, 1] * nb[1] + df[:, 2] * nb[2]))  # (the kernel fuses the multiplies: within the 2^-22 slack)
        keep = keep | ~(c >= thr)
    return keep


def reach_of(verts, faces, eye):
    """(centre, q) as render_frames_impl derives them (mgpu_api.hip) from the vertices the faces use and the camera: rays whose
    origin lies within q of centre consult hints."""
    pts = verts[np.unique(faces)]
    lo, hi = pts.min(0), pts.max(0)
    c = 0.5 * lo + 0.5 * hi
    rho = np.sqrt((np.maximum(hi - c, c - lo) ** 2).sum()) * (1.0 + 2.0 ** -40)
    q = max(np.sqrt(((np.asarray(eye) - c) ** 2).sum()), rho) * (1.0 + 2.0 ** -20)
    return c, q * (1.0 + 2.0 ** -20)


# ---- the scenes --------------------------------------------------------------------------------------------------------------
def _quat_to(dir0):
    """trackball quaternion (x, y, z, w) whose rotation takes the reference camera's -z view direction to dir0"""
    f = np.array([0.0, 0.0, -1.0])
    ax = np.cross(f, dir0)
    s = np.linalg.norm(ax)
    ang = np.arctan2(s, f @ dir0)
    ax = ax / s
    return np.array([ax[0] * np.sin(ang / 2), ax[1] * np.sin(ang / 2), ax[2] * np.sin(ang / 2), np.cos(ang / 2)])


def make_case(S, L, phi, perm=(0, 1, 2), sign=(1, 1, 1), seed=0, W=512, H=8, passes=2, tries=60, mixed=False):
    """One scene.  T = right triangle with legs S (along a coordinate axis) and S sqrt 2 (along a face diagonal), in a plane through
    the eye; the camera looks along that plane at angle phi to the first leg from distance L, and the middle image row's rays (all
    passes) cross the line of the first leg from 2 % of S inside T's corner p1 to several % outside.  The tilt of T against the rays'
    plane is searched (deterministically from `seed`) for the largest number of rays that the reference accepts although they miss
    the round-4 hint box."""
    rng = np.random.default_rng(seed)
    pm = np.zeros((3, 3))
    for i, (p, sg) in enumerate(zip(perm, sign)):
        pm[p, i] = sg
    ex, ey, n0 = pm @ np.array([1.0, 0, 0]), pm @ (np.array([0, 1.0, 1.0]) / np.sqrt(2)), pm @ (np.array([0, -1.0, 1.0]) / np.sqrt(2))
    dir0 = np.cos(phi) * ex + np.sin(phi) * ey
    span_u = float(np.clip(4.0e-4 * L * S, 0.012, 0.3))  # the band the reference's error can reach, in units of S
    fov = float(np.rad2deg(2 * np.arctan(span_u * S * np.sin(phi) / L / W * H / 2)))
    cam = dict(eye=(0.0, 0.0, float(L)), lookat=(0.0, 0.0, 0.0), up=tuple(n0), quat=tuple(_quat_to(dir0)), fov=fov, width=W, height=H)
    frame = O.camera_frame(cam["eye"], cam["lookat"], up=cam["up"], quat=cam["quat"], fov=fov, width=W, height=H)
    row = H // 2
    words_u = rng.integers(0, 2 ** 32, (passes, H, W), dtype=np.uint64)
    words_v = rng.integers(0, 2 ** 32, (passes, H, W), dtype=np.uint64)
    words_v[:, row, :] = 3 << 30  # jitter +0.25 for every pixel of the aimed row: its rays are coplanar (to ~1e-16 rad)
    table = np.zeros((passes, H, W, 4), "<u4")
    for p in range(passes):
        for y in range(H):
            for x in range(W):
                table[p, y, x] = state_for_draws(int(words_u[p, y, x]), int(words_v[p, y, x]))
    org, d = primary_rays(frame, W, words_u[:, row, :], words_v[:, row, :], [row] * passes)
    org, d = org.reshape(-1, 3), d.reshape(-1, 3)
    eye = org[0]
    n = cross(d[0], d[W - 1])
    n = n / np.linalg.norm(n)
    if n @ n0 < 0:
        n = -n
    exp = ex - n * (ex @ n)
    exp = exp / np.linalg.norm(exp)
    eyp = np.cross(n, exp)
    if eyp @ ey < 0:
        eyp = -eyp
    p1c = eye + L * d[W // 2]
    tt = ((p1c - eye) @ eyp) / (d @ eyp)
    xs = (eye[None, :] + tt[:, None] * d) @ exp
    x1 = xs.min() + 0.25 * (xs.max() - xs.min())
    p1 = p1c + (x1 - p1c @ exp) * exp
    p0 = p1 - S * exp
    base = np.stack([p0, p1, p0 + S * np.sqrt(2) * eyp])

    def tilted(psi, v0):  # T rotated by psi about the line v = v0 parallel to its first leg
        return base + (psi * S * np.array([-v0, -v0, 1.0 - v0]))[:, None] * n[None, :]

    # |det| is linear in the tilt: aim it at 1 .. 8 times the reference's threshold
    tri = tilted(1.0e-12, 0.0)
    per_rad = float(np.median(np.abs(triangle_isect(org, d, tri[0], tri[1] - tri[0], tri[2] - tri[0])[4]))) / 1.0e-12
    best = (-1, None)
    for _ in range(tries):
        psi, v0 = EPS1024 * 10 ** rng.uniform(0.0, 0.9) / per_rad, 10 ** rng.uniform(-5, -2)
        tri = tilted(psi, v0)
        ok, _, _, _, _ = triangle_isect(org, d, tri[0], tri[1] - tri[0], tri[2] - tri[0])
        hb = hint_boxes(np.stack([tri[0], tri[1] - tri[0], tri[2] - tri[0]])[None].repeat(2, 0), worth=2.0, rule="r4")
        inbox = hint_keeps(hb[1], org, d)
        score = 1000 * int((ok & ~inbox).sum()) + int(ok.sum())
        if score > best[0]:
            best = (score, tri)
    tri = best[1]
    # four small triangles far behind T, beside the rays' plane: the leaf's "rest"
    far = eye + (L + 6.0 * S + 2.0) * d[W // 2] + 0.05 * S * n
    verts, faces = [tri[0], tri[1], tri[2]], [(0, 1, 2)]
    if mixed:  # a small triangle inside T's box, 60 degrees off T's plane: T's half is no longer planar and takes the padded-box rule
        b = tri[0] + 0.3 * S * (exp + eyp) + 0.02 * S * n
        verts += [b, b + 0.05 * S * exp, b + 0.05 * S * (0.5 * eyp + 0.866 * n)]
        faces.append((3, 4, 5))
    for k in range(4 - (1 if mixed else 0)):
        b = far + 0.02 * S * (k * exp + (k % 2) * eyp)
        i = len(verts)
        verts += [b, b + 0.01 * S * exp, b + 0.01 * S * eyp + 0.003 * S * n]
        faces.append((i, i + 1, i + 2))
    return dict(cam=cam, frame=frame, table=table, verts=np.array(verts, np.float64), faces=np.array(faces, np.uint32), row=row,
                band_org=org, band_dir=d, S=S, L=L, phi=phi, W=W, H=H, passes=passes)


FAMILY = [  # (S, L, phi in degrees, axis permutation, signs): sizes 0.3 .. 10, distances 5 .. 50, six plane orientations
    (3.0, 21.5, 41.0, (0, 1, 2), (1, 1, 1)), (3.0, 21.5, 25.0, (1, 2, 0), (1, -1, 1)), (10.0, 40.0, 30.0, (0, 1, 2), (1, 1, 1)),
    (10.0, 50.0, 69.0, (2, 0, 1), (-1, 1, 1)), (5.0, 30.0, 50.0, (1, 0, 2), (1, 1, -1)), (8.0, 12.0, 35.0, (2, 1, 0), (1, 1, 1)),
    (1.0, 50.0, 45.0, (0, 2, 1), (-1, -1, 1)), (0.3, 25.0, 40.0, (0, 1, 2), (1, 1, 1)), (2.0, 5.0, 60.0, (1, 2, 0), (1, 1, 1)),
    (6.0, 45.0, 20.0, (0, 1, 2), (-1, 1, -1)),
]


def family(W=512, H=8, passes=2):
    return [make_case(S, L, np.deg2rad(phi), perm, sign, seed=100 + i, W=W, H=H, passes=passes, mixed=(i % 3 == 2))
            for i, (S, L, phi, perm, sign) in enumerate(FAMILY)]
