// mgpu_device.hpp -- device-side data layout and the per-ray building blocks of the gfx950 path tracer.
//
// Arithmetic contract: every fp64 expression below keeps the operation ORDER of the reference function it names
// (lighttransport/mallie, paths relative to that tree) and this file is compiled with -ffp-contract=off, so IEEE
// add/sub/mul/div/sqrt give bit-identical results to the x86-64 reference build.  Only acos/sin/cos come from the
// device math library and may differ from glibc in the last ulp (see DESIGN.md "Numerics").
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mgpu.h"

#ifndef MGPU_SAMPLE_MATH
#define MGPU_SAMPLE_MATH 1 // 0 = acos/sincos transcription, 1 = algebraically reduced evaluation (default)
#endif

// ---- what the wave64 emulator of the tests (tests/emu, -DMGPU_EMU) needs spelled differently; on the GPU these ARE the statements they name ----
#ifndef MGPU_EMU
// no-op statements that pin loaded values to registers where they stand (keeps loads of one record together)
#define MGPU_KEEP1(a) asm volatile("" : "+v"(a))
#define MGPU_KEEP2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define MGPU_KEEP4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define MGPU_XCC_ID(v) asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)) // which XCD this wave runs on
#define MGPU_DYN_SHARED(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
// The lanes (of those executing this code) whose predicate holds, as a 64-bit mask.  HIP's __ballot() goes through an integer compare: the
// predicate -- itself a comparison's lane mask -- is turned into 0 / 1 per lane (v_cndmask) and compared with 0 again (v_cmp_ne), two vector
// instructions per ballot; the builtin takes the mask as it is.  (ISA interpreter, C2: 0.7 of 36.3 VALU wave-instructions per ray went there.)
#define MGPU_BALLOT(p) __builtin_amdgcn_ballot_w64(p)
// Does ANY lane executing this -- possibly divergent -- code satisfy p?  Only for choices between two forms that give every lane
// the same bits (a shorter instruction sequence when all lanes qualify); the emulator lets every lane answer for itself.
#define MGPU_ANY(p) (MGPU_BALLOT(p) != 0ull)
// A word every lane of the wave loads from the same address in one instruction -- one value for the wave, whoever writes it meanwhile.
// (The emulator's lanes load one after the other: it takes the first lane's.)
#define MGPU_WAVE_LOAD(x) (x)
#else
#define MGPU_KEEP1(a) ((void)0)
#define MGPU_KEEP2(a, b) ((void)0)
#define MGPU_KEEP4(a, b, c, d) ((void)0)
#define MGPU_XCC_ID(v) ((v) = emu::g_cur->block_idx.x)
#define MGPU_DYN_SHARED(T, name) T *name = reinterpret_cast<T *>(emu::g_cur->dyn_shared)
#define MGPU_ANY(p) (p)
#define MGPU_BALLOT(p) __ballot(p)
#define MGPU_WAVE_LOAD(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#endif

#include "mgpu_sincos.hpp"

namespace mgpu {

constexpr uint32_t kNoMaterial = 0xFFFFFFFFu;
constexpr uint32_t kNoHit = 0xFFFFFFFFu;
constexpr double kDblMax = 1.7976931348623157e+308;
constexpr double kDblEps1024 = 2.220446049250313e-16 * 1024; // std::numeric_limits<double>::epsilon() * 1024

// One triangle per BVH leaf SLOT (position in BVHAccel::indices_), pre-gathered so a leaf is a contiguous run and
// the two index indirections of TestLeafNode (bvh_accel.cc:660-678) disappear.  e1/e2 are the reference's
// p1-p0 / p2-p0 (bvh_accel.cc:606-607) evaluated once on the host: the same IEEE subtraction, the same bits.
struct alignas(16) DTri {
  double p0[3];
  double e1[3];
  double e2[3];
  uint32_t face; // original face index (Intersection::faceID)
  uint32_t mat;  // Mesh::materialIDs[face] or kNoMaterial
};
static_assert(sizeof(DTri) == 80, "DTri must be 80 bytes (5 x 16B loads)");

// One record per INTERIOR node of the reference tree, indexed by that node's index in BVHAccel::nodes_ (records of leaf
// indices stay unused), plus a "super root" at index nn whose child 0 is the root: the boxes of BOTH children and what a
// traversal needs to enter either of them, so that one 128-byte fetch decides both (see wide_node_step).
//   box k   = child k's bmin[3], bmax[3] (the child node's first 48 bytes, verbatim)
//   ref k   = interior child: its node index;  leaf child: first slot of its triangle run (BVHNode::data[1])
//   tag k   = interior child: kWInterior;      leaf child: triangle count (BVHNode::data[0], < kWInterior)
//   tag 0 additionally carries this node's split axis in its top two bits
struct alignas(16) WNode {
  double box0[6];
  double box1[6];
  uint32_t ref0, tag0, ref1, tag1;
  uint32_t pad_[4];
};
static_assert(sizeof(WNode) == 128, "WNode must be 128 bytes (one L2 line)");
constexpr uint32_t kWInterior = 0x3FFFFFFFu;
constexpr uint32_t kWNone = 0xFFFFFFFFu;
// Treelet (k_render_sm with the BVH in HBM): copies of the records of the interior nodes a ray is most likely to enter -- the top
// of the tree, chosen by box surface area when the scene is created -- held in LDS beside the far-child stacks.  A record
// reference with this bit set is an index into that table; inside the table, references to children that are in the table
// too carry the bit, all others are the node indices they always were (the table is closed under "parent of": once a walk
// has left it, it never comes back).  Same records, same decisions, same counts; only where the bytes come from changes.
constexpr uint32_t kWTreelet = 0x80000000u;

struct DScene {
  const MgpuNode *nodes;     // reference 64-byte layout, uploaded verbatim
  const WNode *wnodes;       // nn + 1 wide records (k_wide_layout); record nn is the super root
  uint32_t wroot;            // = nn
  const WNode *treelet;      // treelet_n records, [0] = the super root (see kWTreelet); null when the scene has none
  uint32_t treelet_n;
  uint4 *wstack_overflow;    // wide traversal: far-child stack entries beyond the LDS part, per hardware lane slot
  uint32_t woverflow_cap;    // entries per lane (0: the tree is shallow enough for the LDS part alone)
  const DTri *tris;          // nf entries, slot order
  const double *slot_normal; // slot order: 9 doubles (facevarying n0,n1,n2) or 3 doubles (geometric normal)
  const double *mat_diffuse; // 3*nm
  uint32_t nm;
  int has_fv_normals;
  // original arrays, used by the batched Scene::Trace entry point to fill the full Intersection record
  const double *verts;
  const uint32_t *faces;
  const double *fv_normals; // may be null
  const double *fv_uvs;     // may be null
  // traversal stack overflow area (entries beyond the LDS part), per hardware lane slot
  uint32_t *stack_overflow;
  uint32_t overflow_cap; // entries per lane
  // every reachable node has bmin <= bmax on all axes (checked at scene creation): rays without infinities may then
  // take the min/max form of the slab test, see slab_hit
  int boxes_ordered;
  // every material has three equal diffuse channels (or there are none: the default material is 0.5 grey)
  int grey;
};

struct Hit {
  double t, u, v;
  uint32_t slot; // kNoHit when nothing was hit
};

struct Counters {
  uint32_t rays, nodes, tris;
#ifdef MGPU_UTIL
  uint32_t node_steps, node_lanes, tri_steps, tri_lanes; // per-wave step counts (lane 0 meaningful) and active-lane sums
#endif
};

struct V3 {
  double x, y, z;
};

__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 scale(V3 a, double f) { return v3(a.x * f, a.y * f, a.z * f); }
__device__ __forceinline__ V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
// common.h:66-76
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// ---- IEEE sqrt / reciprocal without the range handling ----------------------------------------------------------
// hipcc expands fp64 sqrt into 18 and 1.0 / x into 11 instructions; 8 resp. 4 of them scale operands near the ends of
// the exponent range and patch 0 / inf / NaN.  These are the same sequences without those steps: bit-identical results
// for operands well inside the range (checked on 2 x 10^10 operands, profiles/microbench/exact_cores.hip), garbage
// outside it -- every caller guards the range and falls back to the plain operator.
__device__ __forceinline__ double sqrt_core(double x) { // 2^-500 < x < 2^500
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
__device__ __forceinline__ double rcp_core(double x) { // 2^-500 < |x| < 2^500
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, e, y);
}

// real3::normalize, common.h:46-56
__device__ __forceinline__ V3 normalized(V3 a) {
  double len = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  if (fabs(len) > 1.0e-6) {
    double inv = 1.0 / len;
    a.x *= inv;
    a.y *= inv;
    a.z *= inv;
  }
  return a;
}

// normalized() as a wave executes it: the short sequences when every active lane's squared length is well inside their
// range and above the reference's 1e-6 guard on the length (then the guard passes and the values are the literal
// form's), the literal form for all lanes otherwise.  19 instead of 30 instructions.
__device__ __forceinline__ V3 normalized_w(V3 a) {
  const double d2 = a.x * a.x + a.y * a.y + a.z * a.z;
  const bool ok = d2 > 1.0e-11 && d2 < 1.0e+100;
  if (__builtin_expect(MGPU_ANY(!ok), 0)) return normalized(a);
  const double inv = rcp_core(sqrt_core(d2));
  return v3(a.x * inv, a.y * inv, a.z * inv);
}
// invDir of BVHAccel::Traverse (bvh_accel.cc:774-802: 1.0 / dir, no zero guard) as a wave executes it.  Returns whether
// all three inverses of this lane came out with 2^-400 < |1/d| < 2^400 -- which implies that the operands were inside
// rcp_core's range, and is the inverse-direction half of a "plain" ray (slab_hit).  One lane outside sends the wave through the
// plain operator.
__device__ __forceinline__ bool inverse_dir_w(V3 dir, double &ix, double &iy, double &iz) {
  ix = rcp_core(dir.x);
  iy = rcp_core(dir.y);
  iz = rcp_core(dir.z);
  const double lo = 0x1p-400, hi = 0x1p+400;
  const bool ok = fabs(ix) > lo && fabs(ix) < hi && fabs(iy) > lo && fabs(iy) < hi && fabs(iz) > lo && fabs(iz) < hi;
  if (__builtin_expect(MGPU_ANY(!ok), 0)) {
    ix = 1.0 / dir.x;
    iy = 1.0 / dir.y;
    iz = 1.0 / dir.z;
  }
  return ok;
}

// 1.0 / det of TriangleIsect (bvh_accel.cc:606) for a det that passed its epsilon test (|det| >= 2^-42, or NaN) as a wave
// executes it: rcp_core when every active lane's |det| is below 2^400, the plain operator for all lanes otherwise
// (a NaN fails the comparison and lands there too, so that its payload is the division's).
__device__ __forceinline__ double inv_det_w(double det) {
  if (__builtin_expect(MGPU_ANY(!(fabs(det) < 0x1p+400)), 0)) return 1.0 / det;
  return rcp_core(det);
}

// ---- slab test ----------------------------------------------------------------------------------------------------
// IntersectRayAABB (bvh_accel.cc:550-593) of the box {b0 = (min.x, min.y), b1 = (min.z, max.x), b2 = (max.y, max.z)}
// (the node's first 48 bytes as three 16-byte loads) against a ray with inverse direction (ix, iy, iz) and direction
// signs (sx, sy, sz), current best t `bt`.
// kPlain = false is the literal form: near / far plane picked by the direction sign, `(a > b) ? a : b` selects (they
// keep the SECOND operand when a NaN is involved, which v_max_f64 would not).
// kPlain = true is for plain rays (inverse_dir_w() true and origin_is_finite()) in a tree with DScene::boxes_ordered: no product can be a NaN, and
// with bmin <= bmax and a finite non-zero inverse the near plane's product is the smaller of the two (rounding is
// monotonic), so the selects collapse to v_min_f64 / v_max_f64.  The values equal the literal form's up to the sign
// of a zero, which none of the three comparisons can see: 25 instead of 39 VALU instructions per box.
template <bool kPlain>
__device__ __forceinline__ bool slab_t(double2 b0, double2 b1, double2 b2, V3 org, double ix, double iy, double iz,
                                       bool sx, bool sy, bool sz, double bt, double &tmin) {
  double tmax;
  if (kPlain) {
    const double lx = (b0.x - org.x) * ix, hx = (b1.y - org.x) * ix;
    const double ly = (b0.y - org.y) * iy, hy = (b2.x - org.y) * iy;
    const double lz = (b1.x - org.z) * iz, hz = (b2.y - org.z) * iz;
    tmin = __builtin_fmax(__builtin_fmax(__builtin_fmin(lx, hx), __builtin_fmin(ly, hy)), __builtin_fmin(lz, hz));
    tmax = __builtin_fmin(__builtin_fmin(__builtin_fmax(lx, hx), __builtin_fmax(ly, hy)), __builtin_fmax(lz, hz));
  } else {
    const double nx = sx ? b1.y : b0.x, fx = sx ? b0.x : b1.y;
    const double ny = sy ? b2.x : b0.y, fy = sy ? b0.y : b2.x;
    const double nz = sz ? b2.y : b1.x, fz = sz ? b1.x : b2.y;
    const double tmin_x = (nx - org.x) * ix, tmax_x = (fx - org.x) * ix;
    const double tmin_y = (ny - org.y) * iy, tmax_y = (fy - org.y) * iy;
    tmin = (tmin_x > tmin_y) ? tmin_x : tmin_y;
    tmax = (tmax_x < tmax_y) ? tmax_x : tmax_y;
    const double tmin_z = (nz - org.z) * iz, tmax_z = (fz - org.z) * iz;
    tmin = (tmin > tmin_z) ? tmin : tmin_z;
    tmax = (tmax < tmax_z) ? tmax : tmax_z;
  }
  return (tmax > 0.0) && (tmin <= tmax) && (tmin <= bt);
}
template <bool kPlain>
__device__ __forceinline__ bool slab_hit(double2 b0, double2 b1, double2 b2, V3 org, double ix, double iy, double iz,
                                         bool sx, bool sy, bool sz, double bt) {
  double tmin;
  return slab_t<kPlain>(b0, b1, b2, org, ix, iy, iz, sx, sy, sz, bt, tmin);
}
// A ray is "plain" when inverse_dir_w() returned true (1/d finite, non-zero, far from the ends of the exponent range on
// every axis) and its origin is finite: (b - o) * inv is then never NaN (an overflowed difference gives +-inf), and
// sign(inv) is the direction sign the literal form selects by.
__device__ __forceinline__ bool origin_is_finite(V3 org) {
  return __builtin_isfinite(org.x) && __builtin_isfinite(org.y) && __builtin_isfinite(org.z);
}

// ---- RNG ----------------------------------------------------------------------------------------------------------
struct Rng {
  uint32_t x, y, z, w;
};
// the 32-bit word randomreal() scales (render.cc:137-168)
__device__ __forceinline__ uint32_t rng_next_u32(Rng &r) {
  uint32_t t = r.x ^ (r.x << 11);
  r.x = r.y;
  r.y = r.z;
  r.z = r.w;
  r.w = (r.w ^ (r.w >> 19)) ^ (t ^ (t >> 8));
  return r.w;
}
// randomreal(), render.cc:137-168
__device__ __forceinline__ double rng_next(Rng &r) {
  uint32_t t = r.x ^ (r.x << 11);
  r.x = r.y;
  r.y = r.z;
  r.z = r.w;
  r.w = (r.w ^ (r.w >> 19)) ^ (t ^ (t >> 8));
  return r.w * (1.0 / 4294967296.0);
}

__host__ __device__ __forceinline__ uint64_t splitmix64_mix(uint64_t x) {
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
// MGPU_RNG_HASH start state; the oracle's mo_hash_state is the same function written independently.
__host__ __device__ __forceinline__ void hash_state(uint64_t seed, uint32_t pass, uint32_t pixel, uint32_t st[4]) {
  const uint64_t golden = 0x9E3779B97F4A7C15ULL;
  uint64_t ctr = seed * golden + (((uint64_t)pass << 32) | (uint64_t)pixel);
  uint64_t a = splitmix64_mix(ctr + golden);
  uint64_t b = splitmix64_mix(ctr + 2 * golden);
  st[0] = (uint32_t)a;
  st[1] = (uint32_t)(a >> 32);
  st[2] = (uint32_t)b;
  st[3] = (uint32_t)(b >> 32);
  if ((st[0] | st[1] | st[2] | st[3]) == 0) st[0] = 1;
}

// ---- traversal stack: first CAP entries in LDS (entry-major, lane-minor: conflict-free), rest in HBM ---------------
// OVF = false: the tree is at most CAP-1 deep, every access is a plain ds_read/ds_write_b32 (the mixed form makes hipcc
// select between an LDS and an HBM ADDRESS and issue a flat_load, which waits on both memory counters).
template <int CAP, bool OVF = true> struct Stack {
  uint32_t *lds;      // &s_stack[wave][0][lane]
  uint32_t *overflow; // this lane's overflow column (null when !OVF)
  __device__ __forceinline__ void put(int i, uint32_t v) const {
    if (!OVF || i < CAP) lds[i * 64] = v;
    else overflow[i - CAP] = v;
  }
  __device__ __forceinline__ uint32_t get(int i) const {
    if (!OVF) return lds[i * 64];
    uint32_t v;
    if (i < CAP) v = lds[i * 64];
    else v = overflow[i - CAP];
    return v;
  }
};

// ---- leaf hints ------------------------------------------------------------------------------------------------------
// TestLeafNode (bvh_accel.cc:640-697) runs TriangleIsect on every triangle of a leaf whose box the ray hits -- up to 15 of them,
// 7.9 per ray on cornellbox_suzanne, one or none of which is hit.  For a leaf of some size the run [first, first + n) of its
// triangles is split at the m that minimises area(box of the first m) * m + area(box of the rest) * (n - m) -- the ORDER of the
// run is the reference's and stays -- and both boxes are kept, padded and rounded outward to float, with m and a cone around each
// half's triangle normals (kHintFloats floats).  When a lane's TRI work on the leaf starts it tests its ray against the two boxes
// once (the kernel's own slab test, in double) and drops the part of the run whose box the ray misses or enters beyond the best
// t -- provided the ray is not grazing that part's triangles, see below: dropping a prefix or a suffix leaves the order of the
// remaining tests -- hence every `t > tBest` decision and the tie rule -- as it was.  Dropped triangles are booked as the tests the
// reference makes.
//
// Why a dropped triangle cannot be one TriangleIsect ACCEPTS (bvh_accel.cc:595-638) -- the rule, derived.  Notation: u = 2^-53 (one
// rounding), s = org - p0, E = |e1| |e2|, Euclidean norms, |a| x+ |b| = the cross product of the absolute values with plus signs
// (its norm is at most 2 / sqrt 3 |a| |b|, attained at a = b = (1, 1, 1)); hats are the values the reference computes, without FMA:
//   p^ = fl(d x e2), det^ = fl(e1 . p^), s^ = fl(org - p0), q^ = fl(s^ x e1),
//   u^ = fl(nu / det^), v^ = fl(nv / det^), t^ = fl(nt / det^) with nu = fl(s^ . p^), nv = fl(q^ . d), nt = fl(e2 . q^),
// accepted iff |det^| >= 1024 eps = 2048 u, 0 <= u^, v^, u^ + v^ <= 1 and 0 <= t^ <= best t.  The exact identity behind the test is
// s det = (s . p) e1 + (q . d) e2 - (e2 . q) d with p = d x e2, q = s x e1, det = e1 . p.  Put X = org + t^ d (exact arithmetic on the
// computed t^) and Y = p0 + u^ e1 + v^ e2; then
//   (X - Y) det^ = s (det^ - det) - (nu - s . p) e1 - (nv - q . d) e2 + (nt - e2 . q) d + (<= 3 u of each term of Y and X: the divisions).
// A three-term dot product fl((a1 b1 + a2 b2) + a3 b3) with result r is off by <= u (1.5 |a| |b| + 2.5 |r|): three products, the
// inner sum (= r - a3 b3, and |a3 b3| <= (sum |ai bi| + |r|) / 2), the outer sum.  All four results here are at most |det^| times
// 1, u^, v^, t^, so their 2.5 |r| parts are a few u of |s|, |e1|, |e2|, t^ |d| after the division -- booked under (*) below.  Then:
//   |p^ - p|     <= u (|d| x+ |e2| + |p|)                                   <= (2 / sqrt 3 + 1) u |d| |e2|  = 2.155 u |d| |e2|
//   |det^ - det| <= 1.5 u |e1| |p^| + |e1| |p^ - p|                          <= 3.655 u |d| E
//   |nu - s . p| <= 1.5 u |s^| |p^| + u |s| |p^| (s^ = s (1 + delta), by component) + |s| |p^ - p|    <= 4.655 u |s| |d| |e2|
//   |q^ - q|     <= u (2 |s| x+ |e1| + |q|)  (each product carries s^'s rounding and its own)        <= 3.31 u |s| |e1|
//   |nv - q . d| <= 1.5 u |q^| |d| + |q^ - q| |d|                                                     <= 4.81 u |s| |e1| |d|
//   |nt - e2 . q| likewise                                                                            <= 4.81 u |s| E
// so, for every test the reference ACCEPTS,
//   |X - Y| <= 17.93 u |s| |d| E / |det^|  +  (*) <= ~20 u (|s| + |e1| + |e2| + t^ |d|).                                    (B)
// With the ONLY guard the reference has, the absolute |det^| >= 2048 u, that is 8.76 / 1024 |s| |d| E: it grows with the distance
// of the origin and with the square of the triangle's size, and it is attained in order of magnitude (a ray within 1e-12 rad of
// a triangle's plane really is accepted ~1 / 1024 |s| |d| E outside the triangle: the round-4 review built one,
// tests/test_hint_soundness_cpu.py measures 0.9 / 1024 on 10^6 accepted near-coplanar tests, tests/hint_family.py aims 10^4 of the
// kernel's own primary rays at that band).  A pad of that size around every half makes the hints worthless on cornellbox_suzanne
// (measured: 5.46 ms with it, 5.42 without hints, 5.12 with the unsound round-4 pad).  So the pad stays small and the rule gets a
// second clause that keeps (B) small instead:
//   a half is dropped iff  the ray misses its box padded by  pad = 2^-8 ext + slack, slack >= 2^-40 (reach + largest |coordinate|),
//                     and  |d . nbar| >= thr   (evaluated in float)
// where (nbar, thr) is a cone around the unit normals n_k = +-(e1 x e2) / |e1 x e2| of the half's triangles -- rho = max |n_k - nbar|
// -- sized so that the second clause implies, for EVERY triangle of the half,
//   |det^_k| >= |e1 x e2| (|d . nbar| - |d| rho) - 3.655 u |d| E_k >= Dsafe := 17.93 u reach E_half / (2^-8 ext),
// i.e. thr = [rho + max_k (Dsafe + 4 u E_k) / |e1 x e2|_k] (1 + 2^-9) + 2^-21 (|d| <= 1 + 2^-10; the float evaluation of d . nbar is within
// 2^-22 of the exact one; nbar is the float vector the record holds, rho is measured against exactly that vector).  Then (B) gives
// |X - Y| <= 2^-8 ext + (*) for any test of the half the reference accepts, X lies inside the padded box with the slack the slab
// test's own roundings need (each of its products is off by <= 3 u |b - org| |1/d|, the slack of a point mu inside a slab is
// mu |1/d|: 2^-40 against 3 u), at a parameter 0 <= t^ <= best t -- and all three clauses of the slab test (tmax > 0, tmin <= tmax,
// tmin <= best t) are true.  So a half whose box FAILS the test while the cone clause holds has no triangle the reference accepts.
// Rays grazing a half's triangles (inside its cone band) keep the half and test it in full; triangles that can never pass the
// determinant guard (|e1 x e2| (1 + 2^-9) + 6 u E_k < 2048 u: zero-area ones) stay out of the cone.
// What bounds |s| |d| by `reach`: a ray consults hints only when it is plain (slab_t), its origin satisfies |org - c| <= Q, where c is
// the centre and rho_s the half diagonal of the box of the scene's vertices and Q = max(|eye - c|, rho_s) for the launch's camera
// (camera rays and every bounce that starts inside the scene's box qualify; a bounce off the far ground plane does not, and tests
// the whole run), and |d| <= 1 + 2^-10, which holds for every ray the kernel makes when the scene's shading normals are no longer
// than 1 + 2^-11 (checked when the scene is created; else the scene gets no hints): reach = (Q + rho_s)(1 + 2^-9).
// tri(k) -> pointer to the 9 doubles p0, e1, e2 of the run's k-th triangle.  rec: kHintFloats floats = box A lo / hi, box B lo / hi
// (12), cone A (nbar, thr), cone B, m as bits, 3 unused (the order leaf_hint_make writes; leaf_hint_pack interleaves the boxes for the
// consultation).  Returns false when the best split saves less than (1 - worth) of the
// expected tests (then it is not worth its two box tests).
constexpr int kHintFloats = 24;
#ifndef MGPU_HINT_CONE_MAX
#define MGPU_HINT_CONE_MAX 0.2 // a half keeps the small pad + cone rule when its cone's threshold is at most this (|cos| of the grazing band it gives up)
#endif
template <typename TriFn>
__device__ __forceinline__ bool leaf_hint_make(TriFn tri, uint32_t n, double worth, const double *c, double q, float *rec) {
  auto grow = [&](uint32_t k, double *lo, double *hi) { // += the k-th triangle: p0, p0 + e1, p0 + e2
    const double *tp = tri(k);
    for (int a = 0; a < 3; ++a) {
      const double p = tp[a], q = tp[a] + tp[3 + a], r = tp[a] + tp[6 + a];
      lo[a] = fmin(lo[a], fmin(p, fmin(q, r)));
      hi[a] = fmax(hi[a], fmax(p, fmax(q, r)));
    }
  };
  auto half_area = [](const double *lo, const double *hi) {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
  };
  const double inf = __builtin_inf();
  double best = inf, whole = 0.0;
  uint32_t best_m = 0u;
  double alo[3] = {inf, inf, inf}, ahi[3] = {-inf, -inf, -inf};
  for (uint32_t m = 1; m <= n; ++m) {
    grow(m - 1, alo, ahi);
    if (m == n) {
      whole = half_area(alo, ahi) * (double)n;
      break;
    }
    double blo[3] = {inf, inf, inf}, bhi[3] = {-inf, -inf, -inf};
    for (uint32_t k = m; k < n; ++k) grow(k, blo, bhi);
    const double c = half_area(alo, ahi) * (double)m + half_area(blo, bhi) * (double)(n - m);
    if (c < best) {
      best = c;
      best_m = m;
    }
  }
  if (best_m == 0u || !(best < worth * whole)) return false; // (NaN boxes: no hint)
  const double u = 0x1p-53, up = 1.0 + 0x1p-9; // up: |d| <= 1 + 2^-10 and the roundings of this function
  for (int part = 0; part < 2; ++part) {
    const uint32_t k0 = part ? best_m : 0u, k1 = part ? n : best_m;
    double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    double e_half = 0.0, sum[3] = {0.0, 0.0, 0.0};
    // the triangle's e1 x e2, its length and E_k; `live`: it can pass the determinant guard at all
    auto normal = [&](uint32_t k, double *nk, double &len, double &ek) -> bool {
      const double *tp = tri(k);
      nk[0] = tp[4] * tp[8] - tp[5] * tp[7];
      nk[1] = tp[5] * tp[6] - tp[3] * tp[8];
      nk[2] = tp[3] * tp[7] - tp[4] * tp[6];
      len = sqrt(nk[0] * nk[0] + nk[1] * nk[1] + nk[2] * nk[2]);
      ek = sqrt(tp[3] * tp[3] + tp[4] * tp[4] + tp[5] * tp[5]) * sqrt(tp[6] * tp[6] + tp[7] * tp[7] + tp[8] * tp[8]) * up;
      return !(len * up + 6.0 * u * ek < 2048.0 * u);
    };
    for (uint32_t k = k0; k < k1; ++k) {
      grow(k, lo, hi);
      double nk[3], len, ek;
      if (!normal(k, nk, len, ek)) continue;
      e_half = fmax(e_half, ek);
      if (len > 0.0) {
        const double sg = (nk[0] * sum[0] + nk[1] * sum[1] + nk[2] * sum[2]) < 0.0 ? -1.0 : 1.0;
        for (int a = 0; a < 3; ++a) sum[a] += sg * nk[a] / len;
      }
    }
    double ext = 0.0, big = 0.0;
    for (int a = 0; a < 3; ++a) {
      ext = fmax(ext, hi[a] - lo[a]);
      big = fmax(big, fmax(fabs(lo[a]), fabs(hi[a])));
    }
    // reach of THIS half: |org - p0| |d| <= (q + the farthest corner of the half's box from c) (1 + 2^-9) for every consulting ray
    double far2 = 0.0;
    for (int a = 0; a < 3; ++a) {
      const double f = fmax(fabs(lo[a] - c[a]), fabs(hi[a] - c[a]));
      far2 += f * f;
    }
    const double reach = (q + sqrt(far2)) * up;
    const double cbig = fmax(fabs(c[0]), fmax(fabs(c[1]), fabs(c[2])));
    const double pad_geo = ext * 0x1p-8, slack = 0x1p-20 * (cbig + 2.0 * reach + big); // (2^-40 (reach + big) for the double form)
    // the cone: nbar = the float rounding of the normalised sum of the (sign-aligned) unit normals; rho and thr against that vector
    const double sl = sqrt(sum[0] * sum[0] + sum[1] * sum[1] + sum[2] * sum[2]);
    float nb[3] = {0.0f, 0.0f, 0.0f};
    if (sl > 0.0)
      for (int a = 0; a < 3; ++a) nb[a] = (float)(sum[a] / sl);
    const double d_safe = 17.93 * u * reach * e_half / pad_geo; // (inf or NaN for a half without extent: never safe, see below)
    double thr = 0.0; // a half none of whose triangles is live can always be dropped
    for (uint32_t k = k0; k < k1; ++k) {
      double nk[3], len, ek;
      if (!normal(k, nk, len, ek)) continue;
      double dm = 0.0, dp = 0.0;
      for (int a = 0; a < 3; ++a) {
        const double x = nk[a] / len;
        dm += (x - (double)nb[a]) * (x - (double)nb[a]);
        dp += (x + (double)nb[a]) * (x + (double)nb[a]);
      }
      const double t = sqrt(fmin(dm, dp)) * up + (d_safe + 4.0 * u * ek) / len * up + 0x1p-21;
      thr = (t > thr) ? t : ((t == t) ? thr : inf); // NaN (len == 0, no extent): never safe
    }
    // Which of the two sound rules this half gets: a tight cone (its triangles share a plane, or nearly: the Cornell walls, a floor
    // triangle that the builder put into a leaf of Suzanne's) keeps the small pad; any other half takes the pad that covers (B) at
    // the determinant guard, kappa E_half reach with kappa = 9 / 1024 > 8.76 / 1024, and needs no cone (thr = 0: always passes).
    double pad = pad_geo + slack;
    float thr_f = __double2float_ru(thr);
    if (!(thr <= (double)MGPU_HINT_CONE_MAX)) {
      pad = (9.0 / 1024.0) * e_half * reach + slack;
      thr_f = 0.0f;
    }
    if (!(pad < inf)) return false;
    for (int a = 0; a < 3; ++a) {
      rec[6 * part + a] = __double2float_rd(lo[a] - pad);
      rec[6 * part + 3 + a] = __double2float_ru(hi[a] + pad);
      // leaf_hint_apply multiplies (box coordinate - origin) by 1 / d in FLOAT: with |1 / d| < 2^100 and an origin below 2^26 (the host
      // gives a launch hints only when |c| + Q is: mgpu_api.hip) a coordinate below 2^26 keeps every product finite.  A padded box that
      // reaches farther gets no record (an infinite product could turn a hit half into a missed one).
      if (!(fabsf(rec[6 * part + a]) < 0x1p26f && fabsf(rec[6 * part + 3 + a]) < 0x1p26f)) return false;
    }
    rec[12 + 4 * part + 0] = nb[0];
    rec[12 + 4 * part + 1] = nb[1];
    rec[12 + 4 * part + 2] = nb[2];
    rec[12 + 4 * part + 3] = thr_f;
  }
  rec[20] = __uint_as_float(best_m);
  rec[21] = rec[22] = rec[23] = 0.0f;
  return true;
}
// The consultation: [tri_cur, tri_end) = the leaf's whole run on entry, what is left of it on return; returns the number of
// triangles dropped.  f0 .. f2 = the record's boxes, cA / cB = its cones, m = its split.  For a ray that may consult hints only.
// Evaluated in FLOAT (the boxes are floats; the ray's origin, inverse direction and direction are rounded to nearest, the best t
// upwards): every float slab product is the exact one of a plane shifted by <= 2^-24 |org| and is off by <= 3.1 x 2^-24 of itself,
// which the 2^-20 (|c| + 2 reach + |coordinate|) that leaf_hint_make adds to every pad turns into slack on the right side of all three
// clauses; |1 / d| < 2^100 (part of the ray's permission), box coordinates below 2^26 (leaf_hint_make refuses any other record) and
// origins below 2^26 (|c| + Q is, or the launch has no hints: mgpu_api.hip) keep every product below 2^127, i.e. finite.  Half the issue
// slots of the double form.
// (The cone clauses are evaluated for every consulting lane, not only behind a missed box: six float FMAs against a second,
// dependent trip to LDS in the middle of the step.)
// Both boxes at once: the record's first twelve floats are stored INTERLEAVED -- (A.lo.x, B.lo.x), (A.lo.y, B.lo.y), (A.lo.z, B.lo.z),
// (A.hi.x, B.hi.x), ... (leaf_hint_pack) -- so that the six subtractions and six multiplications of the two slab tests are six v_pk_add_f32 and
// six v_pk_mul_f32 on the pairs as they were loaded (the origin and 1 / d broadcast by op_sel): the same IEEE operations on the same operands as
// the one-box form, half the instructions (tools/isa_profile.py: -0.3 VALU wave-instructions per ray on C2).
typedef float mgpu_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void leaf_hint_pack(const float *rec, float *out) {
  for (int a = 0; a < 3; ++a) {
    out[2 * a] = rec[a];              // A.lo
    out[2 * a + 1] = rec[6 + a];      // B.lo
    out[6 + 2 * a] = rec[3 + a];      // A.hi
    out[6 + 2 * a + 1] = rec[9 + a];  // B.hi
  }
  for (int a = 0; a < 4; ++a) { // the cones likewise: (nbar_A.x, nbar_B.x), (.y, .y), (.z, .z), (thr_A, thr_B)
    out[12 + 2 * a] = rec[12 + a];
    out[12 + 2 * a + 1] = rec[16 + a];
  }
  for (int k = 20; k < kHintFloats; ++k) out[k] = rec[k];
}
__device__ __forceinline__ uint32_t leaf_hint_apply(float4 f0, float4 f1, float4 f2, float4 cA, float4 cB, uint32_t m, V3 org, V3 dir, double ix,
                                                    double iy, double iz, double bt, uint32_t &tri_cur, uint32_t &tri_end) {
  const float dx = (float)dir.x, dy = (float)dir.y, dz = (float)dir.z;
  const float ox = (float)org.x, oy = (float)org.y, oz = (float)org.z, jx = (float)ix, jy = (float)iy, jz = (float)iz;
  const float bt_up = __double2float_ru(bt);
  // cA = (nbar_A.x, nbar_B.x, nbar_A.y, nbar_B.y), cB = (nbar_A.z, nbar_B.z, thr_A, thr_B): d . nbar of both halves in three packed operations
  // (NaN compares false: the half counts as grazed and stays)
  const mgpu_f2 nx = {cA.x, cA.y}, ny = {cA.z, cA.w}, nz = {cB.x, cB.y};
  const mgpu_f2 dn = __builtin_elementwise_fma(nx, (mgpu_f2){dx, dx}, __builtin_elementwise_fma(ny, (mgpu_f2){dy, dy}, nz * dz));
  const bool sA = fabsf(dn[0]) >= cB.z;
  const bool sB = fabsf(dn[1]) >= cB.w;
  // f0 = (A.lo.x, B.lo.x, A.lo.y, B.lo.y), f1 = (A.lo.z, B.lo.z, A.hi.x, B.hi.x), f2 = (A.hi.y, B.hi.y, A.hi.z, B.hi.z)
  const mgpu_f2 lx = {f0.x, f0.y}, ly = {f0.z, f0.w}, lz = {f1.x, f1.y}, hx = {f1.z, f1.w}, hy = {f2.x, f2.y}, hz = {f2.z, f2.w};
  const mgpu_f2 ax = (lx - ox) * jx, bx = (hx - ox) * jx, ay = (ly - oy) * jy, by = (hy - oy) * jy, az = (lz - oz) * jz, bz = (hz - oz) * jz;
  bool hit[2];
  for (int k = 0; k < 2; ++k) {
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax[k], bx[k]), __builtin_fminf(ay[k], by[k])), __builtin_fminf(az[k], bz[k]));
    const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax[k], bx[k]), __builtin_fmaxf(ay[k], by[k])), __builtin_fmaxf(az[k], bz[k]));
    hit[k] = (tmax > 0.0f) && (tmin <= tmax) && (tmin <= bt_up);
  }
  const bool hA = !sA || hit[0];
  const bool hB = !sB || hit[1];
  const uint32_t whole = tri_end - tri_cur, mid = tri_cur + m;
  if (!hB) tri_end = mid;
  if (!hA) tri_cur = hB ? mid : tri_end;
  return whole - (tri_end - tri_cur);
}

// ---- wide traversal (BVH in HBM) ------------------------------------------------------------------------------------
// BVHAccel::Traverse (bvh_accel.cc:805-834) pops a node, tests ITS box against the best t so far, and on a hit pushes the
// far child, then the near child.  A visited node costs one dependent fetch that way, and with the tree in HBM the walk
// is a chain of such fetches.  The wide form fetches an interior node's WNode record instead -- both children's boxes in
// one 128-byte line -- and decides both at once:
//   * the near child is what the reference pops next, with an unchanged best t: it is tested now and, when hit, entered
//     directly (leaf: its triangle run is in the record, the leaf's own node is never fetched);
//   * the far child's test reads `tmax > 0 && tmin <= tmax && tmin <= best`.  Only the last clause can change until the
//     reference pops it, and `best` only shrinks: the child goes on the stack with its exact tmin iff it passes now, and
//     `tmin <= best` is asked again when it is popped.  Same accept / reject decisions, same order.
// The reference pops (and counts) every pushed node; both children of every entered interior node are pushed there, so
// nodes visited = 1 + 2 * (interior nodes entered) -- counted here as 2 per record, less 1 for the super root's dummy.
// The stack holds far children only: at most one per interior level, in practice < 8 (tools/wide_proto.c: 99 % of the
// rays of the 1M-triangle grid never hold more than 6).  K entries of {ref, tag, tmin} per lane live in LDS
// (entry-major, lane-minor: conflict-free), deeper ones in the lane's HBM column.
template <int K> struct WStack {
  __attribute__((address_space(3))) unsigned long long *rt; // &s_rt[wave][0][lane]: ref | tag << 32
  __attribute__((address_space(3))) double *tm; // &s_tm[wave][0][lane]
  uint4 *overflow;                              // this lane's column, entries K.. (null when the tree is shallow)
  __device__ __forceinline__ void put(int i, uint32_t ref, uint32_t tag, double t) const {
    if (i < K) {
      rt[i * 64] = (unsigned long long)ref | ((unsigned long long)tag << 32);
      tm[i * 64] = t;
    } else {
      const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
      overflow[i - K] = make_uint4(ref, tag, (uint32_t)tb, (uint32_t)(tb >> 32));
    }
  }
  __device__ __forceinline__ void get(int i, uint32_t &ref, uint32_t &tag, double &t) const {
    if (i < K) {
      const unsigned long long a = rt[i * 64];
      ref = (uint32_t)a;
      tag = (uint32_t)(a >> 32);
      t = tm[i * 64];
    } else {
      const uint4 a = overflow[i - K];
      ref = a.x;
      tag = a.y;
      t = __longlong_as_double((long long)(((unsigned long long)a.w << 32) | (unsigned long long)a.z));
    }
  }
  // bytes of LDS per wave
  static constexpr size_t kWaveBytes = (size_t)K * 64 * (sizeof(unsigned long long) + sizeof(double));
  __device__ __forceinline__ void bind(unsigned char *wave_base, int lane) {
    rt = (__attribute__((address_space(3))) unsigned long long *)(wave_base) + lane;
    tm = (__attribute__((address_space(3))) double *)(wave_base + (size_t)K * 64 * sizeof(unsigned long long)) + lane;
  }
};

// The same stack with 12-byte entries (k_render_w5, mgpu_render_w5.hip): {ref, tag} in one word.  Leaf child: first slot (< 2^24) |
// triangle count (< 64) << 24; interior child: kPkInterior | its reference (node index < 2^30, or kWTreelet | table index).  The
// host launches that kernel only for scenes whose references fit (mgpu_api.hip, w5_eligible).
constexpr uint32_t kPkInterior = 0x40000000u;
template <int K> struct WStackP {
  __attribute__((address_space(3))) uint32_t *rt; // &s_rt[wave][0][lane]
  __attribute__((address_space(3))) double *tm;   // &s_tm[wave][0][lane]
  uint4 *overflow;                                // this lane's column, entries K.. (null when the tree is shallow)
  __device__ __forceinline__ void put(int i, uint32_t ref, uint32_t tag, double t) const {
    const uint32_t pk = tag == kWInterior ? (ref | kPkInterior) : (ref | (tag << 24));
    if (i < K) {
      rt[i * 64] = pk;
      tm[i * 64] = t;
    } else {
      const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
      overflow[i - K] = make_uint4(pk, 0u, (uint32_t)tb, (uint32_t)(tb >> 32));
    }
  }
  __device__ __forceinline__ void get(int i, uint32_t &ref, uint32_t &tag, double &t) const {
    uint32_t pk;
    if (i < K) {
      pk = rt[i * 64];
      t = tm[i * 64];
    } else {
      const uint4 a = overflow[i - K];
      pk = a.x;
      t = __longlong_as_double((long long)(((unsigned long long)a.w << 32) | (unsigned long long)a.z));
    }
    const bool interior = (pk & kPkInterior) != 0u;
    ref = interior ? (pk & ~kPkInterior) : (pk & 0x00FFFFFFu);
    tag = interior ? kWInterior : (pk >> 24);
  }
  static constexpr size_t kWaveBytes = (size_t)K * 64 * (sizeof(uint32_t) + sizeof(double));
  __device__ __forceinline__ void bind(unsigned char *wave_base, int lane) {
    tm = (__attribute__((address_space(3))) double *)(wave_base) + lane;
    rt = (__attribute__((address_space(3))) uint32_t *)(wave_base + (size_t)K * 64 * sizeof(double)) + lane;
  }
};

enum : int { WT_NODE = 0, WT_TRI = 1, WT_DONE = 2 };

// Up to REPS interior nodes for the calling lane.  `cur` = record to enter next (kWNone: take one from the stack), `sp` =
// entries on the stack.  Returns WT_TRI with [tri_cur, tri_end) set when a leaf was opened, WT_DONE when the stack ran
// empty, WT_NODE when the repetitions are used up.  n_nodes += 2 per record entered.
// TL: references with kWTreelet set are read from the treelet table at `tl` in LDS.
template <bool kPlain, int REPS, int K, bool TL = false, typename STK = WStack<K>>
__device__ __forceinline__ int wide_node_step(const WNode *__restrict__ wn, const STK &stk, V3 org, double ix,
                                              double iy, double iz, bool sx, bool sy, bool sz, uint32_t sgn /* sx | sy << 1 | sz << 2 */,
                                              double bt, uint32_t &cur,
                                              int &sp, uint32_t &tri_cur, uint32_t &tri_end, uint32_t &n_nodes,
                                              const unsigned char *tl = nullptr, const bool plain_rt = kPlain) {
  int res = WT_NODE;
#pragma unroll 1
  for (int rep = 0; rep < REPS; ++rep) {
    if (cur == kWNone) {
      // next far child whose tmin still is <= the best t (bvh_accel.cc:586: `tmin <= maxT`, the only clause that can
      // have changed since it was pushed)
#ifdef MGPU_EXP_POP1 // (experiment: one pop per repetition, a culled entry costs the repetition)
      if (sp == 0) {
        res = WT_DONE;
        break;
      }
      uint32_t ref, tag;
      double tmin;
      --sp;
      stk.get(sp, ref, tag, tmin);
      if (!(tmin <= bt)) continue;
#else
      uint32_t ref = 0, tag = 0;
      bool got = false;
      while (sp > 0) {
        double tmin;
        --sp;
        stk.get(sp, ref, tag, tmin);
        if (tmin <= bt) {
          got = true;
          break;
        }
      }
      if (!got) {
        res = WT_DONE;
        break;
      }
#endif
      if (tag != kWInterior) {
        tri_cur = ref;
        tri_end = ref + tag;
        res = WT_TRI;
        break;
      }
      cur = ref;
    }
    double2 a0, a1, a2, a3, a4, a5;
    uint4 m;
    if (TL && (cur & kWTreelet) != 0u) {
      // an LDS address in its own address space, and an opaque statement in this branch only: otherwise hipcc folds the two
      // branches into ONE flat_load from a selected address (see Stack above), which is neither an LDS nor a global load
      typedef __attribute__((address_space(3))) const unsigned char lds_byte;
      typedef double lds_d2 __attribute__((ext_vector_type(2)));
      typedef uint32_t lds_u4 __attribute__((ext_vector_type(4)));
      lds_byte *r = (lds_byte *)tl + (cur & ~kWTreelet) * (uint32_t)sizeof(WNode);
      const lds_d2 l0 = *(__attribute__((address_space(3))) const lds_d2 *)(r), l1 = *(__attribute__((address_space(3))) const lds_d2 *)(r + 16),
                   l2 = *(__attribute__((address_space(3))) const lds_d2 *)(r + 32), l3 = *(__attribute__((address_space(3))) const lds_d2 *)(r + 48),
                   l4 = *(__attribute__((address_space(3))) const lds_d2 *)(r + 64), l5 = *(__attribute__((address_space(3))) const lds_d2 *)(r + 80);
      const lds_u4 lm = *(__attribute__((address_space(3))) const lds_u4 *)(r + 96);
      a0 = make_double2(l0.x, l0.y); a1 = make_double2(l1.x, l1.y); a2 = make_double2(l2.x, l2.y);
      a3 = make_double2(l3.x, l3.y); a4 = make_double2(l4.x, l4.y); a5 = make_double2(l5.x, l5.y);
      m = make_uint4(lm.x, lm.y, lm.z, lm.w);
      MGPU_KEEP1(m.x);
    } else {
      const WNode *r = wn + cur;
      const double2 *q = reinterpret_cast<const double2 *>(r);
      a0 = q[0]; a1 = q[1]; a2 = q[2]; a3 = q[3]; a4 = q[4]; a5 = q[5];
      m = *reinterpret_cast<const uint4 *>(&r->ref0);
    }
    // all seven loads of the record are issued together (the compiler would sink the last behind the box tests)
    MGPU_KEEP4(m.x, m.y, m.z, m.w);
    n_nodes += 2;
    double t0, t1;
#ifdef MGPU_EXP_ONE_LOOP // (experiment: one record loop for both slab forms, the form picked per repetition by a wave-uniform branch)
    bool h0, h1;
    if (plain_rt) {
      h0 = slab_t<true>(a0, a1, a2, org, ix, iy, iz, sx, sy, sz, bt, t0);
      h1 = slab_t<true>(a3, a4, a5, org, ix, iy, iz, sx, sy, sz, bt, t1);
    } else {
      h0 = slab_t<false>(a0, a1, a2, org, ix, iy, iz, sx, sy, sz, bt, t0);
      h1 = slab_t<false>(a3, a4, a5, org, ix, iy, iz, sx, sy, sz, bt, t1);
    }
#else
    const bool h0 = slab_t<kPlain>(a0, a1, a2, org, ix, iy, iz, sx, sy, sz, bt, t0);
    const bool h1 = slab_t<kPlain>(a3, a4, a5, org, ix, iy, iz, sx, sy, sz, bt, t1);
#endif
    const uint32_t axis = m.y >> 30, tag0 = m.y & kWInterior, tag1 = m.w;
    const bool nearIsSecond = ((sgn >> axis) & 1u) != 0u; // dirSign[node.axis], bvh_accel.cc:818-824
    const bool hn = nearIsSecond ? h1 : h0, hf = nearIsSecond ? h0 : h1;
    const uint32_t refn = nearIsSecond ? m.z : m.x, tagn = nearIsSecond ? tag1 : tag0;
    const uint32_t reff = nearIsSecond ? m.x : m.z, tagf = nearIsSecond ? tag0 : tag1;
    const double tf = nearIsSecond ? t0 : t1;
    if (hf && tagf != 0u) { // an empty leaf (count 0) has nothing to test
      stk.put(sp, reff, tagf, tf);
      ++sp;
    }
    cur = kWNone;
    if (hn) {
      if (tagn == kWInterior) {
        cur = refn;
      } else if (tagn != 0u) {
        tri_cur = refn;
        tri_end = refn + tagn;
        res = WT_TRI;
        break;
      }
    }
  }
  if (res == WT_NODE && cur == kWNone && sp == 0) res = WT_DONE; // nothing left: spare the caller a round that finds out
  return res;
}

// ---- shared leaves: the TRI step of the wave-scheduled kernels with 2^sh lanes per open leaf --------------------------
// With at most 64 >> sh lanes holding an open leaf (mT = their mask, cT = their number), 2^sh lanes work on each: lane L
// serves the (L >> sh)-th open leaf and tests its triangles first + (L & (m - 1)), + m, ... with the OWNER's ray (fetched
// across lanes), idle and NODE / SHADE lanes included -- their own state is untouched.  The m partial results are merged by
// the rule the reference's in-order loop obeys for ordinary numbers -- smallest t, the LATER triangle on equal t
// (TriangleIsect rejects only `t > tBest`, bvh_accel.cc:631) -- and the owner applies the same `t > bt` test to the merged
// candidate.  A NaN t (which that loop would accept, and after which it accepts everything) cannot be merged this way: a
// step that produces one changes nothing and returns false, and the caller redoes it in order with the owners alone.
// Called by the whole wave.  tbl: 64 bytes of LDS of this wave.  `owner`: this lane holds an open leaf [tri_cur, tri_end).
// On true the owners' (bt, bu, bv, bslot) are updated and tri_cur has advanced by up to MAX_TRIPS << sh triangles;
// my_trips = loop trips this lane made as a worker.
template <bool LDS_TRIS, int MAX_TRIPS>
__device__ __forceinline__ bool shared_leaves_step(unsigned long long mT, int cT, int sh, int lane, unsigned char *tbl, bool owner,
                                                   const unsigned char *lds_tris, const DTri *tris, V3 org, V3 dir,
                                                   uint32_t &tri_cur, uint32_t tri_end, double &bt, double &bu, double &bv,
                                                   uint32_t &bslot, uint32_t &n_tris, uint32_t &my_trips) {
  const int m = 1 << sh;
  const uint32_t rank = (uint32_t)__popcll(mT & ((1ull << lane) - 1ull));
  if (owner) tbl[rank] = (unsigned char)lane;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int grp = lane >> sh, sub = lane & (m - 1);
  const bool serving = grp < cT;
  const int own = serving ? (int)tbl[grp] : lane;
  const V3 o = v3(__shfl(org.x, own), __shfl(org.y, own), __shfl(org.z, own));
  const V3 d = v3(__shfl(dir.x, own), __shfl(dir.y, own), __shfl(dir.z, own));
  const uint32_t first = (uint32_t)__shfl((int)tri_cur, own), last = (uint32_t)__shfl((int)tri_end, own);
  double lt = __builtin_inf(), lu = 0.0, lv = 0.0;
  uint32_t ls = kNoHit;
  if (serving) {
    uint32_t i = first + (uint32_t)sub;
#pragma unroll 1
    for (int rep = 0; rep < MAX_TRIPS && i < last; ++rep, i += (uint32_t)m) {
      double2 a0, a1, a2, a3;
      double e2z;
      if (LDS_TRIS) {
        const unsigned char *tp = lds_tris + (size_t)i * 80;
        a0 = *reinterpret_cast<const double2 *>(tp);
        a1 = *reinterpret_cast<const double2 *>(tp + 16);
        a2 = *reinterpret_cast<const double2 *>(tp + 32);
        a3 = *reinterpret_cast<const double2 *>(tp + 48);
        e2z = *reinterpret_cast<const double *>(tp + 64);
      } else {
        const DTri *tp = tris + i;
        a0 = reinterpret_cast<const double2 *>(tp)[0];
        a1 = reinterpret_cast<const double2 *>(tp)[1];
        a2 = reinterpret_cast<const double2 *>(tp)[2];
        a3 = reinterpret_cast<const double2 *>(tp)[3];
        e2z = tp->e2[2];
      }
      ++n_tris;
      // TriangleIsect, bvh_accel.cc:595-638
      const V3 p0 = v3(a0.x, a0.y, a1.x), e1 = v3(a1.y, a2.x, a2.y), e2 = v3(a3.x, a3.y, e2z);
      const V3 p = cross(d, e2);
      const double det = dot(e1, p);
      if (!(fabs(det) < kDblEps1024)) {
        const double invDet = inv_det_w(det);
        const V3 sv = o - p0;
        const V3 q = cross(sv, e1);
        const double u = dot(sv, p) * invDet;
        const double v = dot(q, d) * invDet;
        const double t = dot(e2, q) * invDet;
        const bool rej = (u < 0.0 || u > 1.0) || (v < 0.0 || u + v > 1.0) || (t < 0.0 || t > lt);
        if (!rej) {
          lt = t;
          lu = u;
          lv = v;
          ls = i;
        }
      }
    }
  }
  my_trips = serving ? min(((last - first) + (uint32_t)(m - 1 - sub)) >> sh, (uint32_t)MAX_TRIPS) : 0u;
  if (MGPU_BALLOT(lt != lt) != 0ull) { // a NaN candidate somewhere: nothing is merged, nothing has changed
    n_tris -= my_trips;              // (the caller's in-order pass counts these tests)
    return false;
  }
  for (int x = 1; x < m; x <<= 1) {
    const double pt = __shfl_xor(lt, x), pu = __shfl_xor(lu, x), pv = __shfl_xor(lv, x);
    const uint32_t ps = (uint32_t)__shfl_xor((int)ls, x);
    // the partner wins with a smaller t, or with an equal t and the later triangle (kNoHit never wins: its t is +inf, and an
    // equal +inf from a real triangle loses nothing -- the owner's `t > bt` test rejects it either way)
    const bool take = ps != kNoHit && (ls == kNoHit || pt < lt || (pt == lt && ps > ls));
    if (take) {
      lt = pt;
      lu = pu;
      lv = pv;
      ls = ps;
    }
  }
  const int from = (int)(rank << sh); // the lanes of this owner's group all hold the merged candidate
  const double ct = __shfl(lt, from), cu = __shfl(lu, from), cv = __shfl(lv, from);
  const uint32_t cs = (uint32_t)__shfl((int)ls, from);
  if (owner) {
    if (cs != kNoHit && !(ct > bt)) {
      bt = ct;
      bu = cu;
      bv = cv;
      bslot = cs;
    }
    tri_cur += min(tri_end - tri_cur, (uint32_t)(MAX_TRIPS << sh));
  }
  return true;
}

// ---- BVHAccel::Traverse (bvh_accel.cc:773-844) without the final BuildIntersection ----------------------------------
// while-while form: every lane pops and box-tests nodes until it holds a leaf (or runs dry), then the wave tests leaf
// triangles together.  Pop order, the near/far push order and the in-leaf triangle order are the reference's, so
// exact-t ties resolve identically and node/triangle counts equal the CPU's.
template <int CAP, bool OVF>
__device__ __forceinline__ void traverse(const DScene &sc, const Stack<CAP, OVF> &stk, V3 org, V3 dir, Hit &h, Counters &c) {
  const bool sx = dir.x < 0.0, sy = dir.y < 0.0, sz = dir.z < 0.0;
  const uint32_t sgn = (sx ? 1u : 0u) | (sy ? 2u : 0u) | (sz ? 4u : 0u);
  double ix, iy, iz;
  const bool inv_ok = inverse_dir_w(dir, ix, iy, iz); // 1.0 / dir, no zero guard, as the reference
  // wave-uniform: every active lane's ray may take the min/max form of the box test (slab_hit)
  const bool all_plain = !MGPU_ANY(!(sc.boxes_ordered && inv_ok && origin_is_finite(org)));
  h.t = kDblMax;
  h.u = 0.0;
  h.v = 0.0;
  h.slot = kNoHit;
  int sp = 0;
  stk.put(0, 0u);
  uint32_t leaf_first = 0, leaf_cnt = 0;
  uint32_t nnodes = 0, ntris = 0;
  for (;;) {
    while (sp >= 0 && leaf_cnt == 0) {
#ifdef MGPU_UTIL
      { // one count per wave-level execution of this body: the first active lane books it
        const unsigned long long act = MGPU_BALLOT(1);
        if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)act) - 1)) c.node_steps += 1;
      }
#endif
      const uint32_t ni = stk.get(sp);
      --sp;
      ++nnodes;
      const MgpuNode *nd = sc.nodes + ni;
      const double2 b0 = *reinterpret_cast<const double2 *>(&nd->bmin[0]); // bmin.x bmin.y
      const double2 b1 = *reinterpret_cast<const double2 *>(&nd->bmin[2]); // bmin.z bmax.x
      const double2 b2 = *reinterpret_cast<const double2 *>(&nd->bmax[1]); // bmax.y bmax.z
      int4 meta = *reinterpret_cast<const int4 *>(&nd->flag);              // flag axis data0 data1
      // issued with the three loads above (same 64-byte node), not sunk into the hit branch as a second dependent load
      MGPU_KEEP4(meta.x, meta.y, meta.z, meta.w);
      // IntersectRayAABB, bvh_accel.cc:550-593
      const bool hit = all_plain ? slab_hit<true>(b0, b1, b2, org, ix, iy, iz, sx, sy, sz, h.t)
                                 : slab_hit<false>(b0, b1, b2, org, ix, iy, iz, sx, sy, sz, h.t);
      if (hit) {
        if (meta.x == 0) {
          const bool nearIsSecond = ((sgn >> (uint32_t)meta.y) & 1u) != 0u; // dirSign[node.axis]
          const uint32_t c0 = (uint32_t)meta.z, c1 = (uint32_t)meta.w;
          stk.put(sp + 1, nearIsSecond ? c0 : c1); // far
          stk.put(sp + 2, nearIsSecond ? c1 : c0); // near: popped first
          sp += 2;
        } else {
          leaf_cnt = (uint32_t)meta.z;
          leaf_first = (uint32_t)meta.w;
        }
      }
    }
    if (leaf_cnt == 0) break;
    // TestLeafNode + TriangleIsect, bvh_accel.cc:595-697
    for (uint32_t i = 0; i < leaf_cnt; ++i) {
#ifdef MGPU_UTIL
      {
        const unsigned long long act = MGPU_BALLOT(1);
        if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)act) - 1)) c.tri_steps += 1;
      }
#endif
      const DTri *tp = sc.tris + (leaf_first + i);
      const double2 a0 = reinterpret_cast<const double2 *>(tp)[0]; // p0.x p0.y
      const double2 a1 = reinterpret_cast<const double2 *>(tp)[1]; // p0.z e1.x
      const double2 a2 = reinterpret_cast<const double2 *>(tp)[2]; // e1.y e1.z
      const double2 a3 = reinterpret_cast<const double2 *>(tp)[3]; // e2.x e2.y
      const double e2z = tp->e2[2];
      ++ntris;
      const V3 p0 = v3(a0.x, a0.y, a1.x), e1 = v3(a1.y, a2.x, a2.y), e2 = v3(a3.x, a3.y, e2z);
      const V3 p = cross(dir, e2);
      const double det = dot(e1, p);
      if (fabs(det) < kDblEps1024) continue;
      const double invDet = inv_det_w(det); // 1.0 / det
      const V3 s = org - p0;
      const V3 q = cross(s, e1);
      const double u = dot(s, p) * invDet;
      const double v = dot(q, dir) * invDet;
      const double t = dot(e2, q) * invDet;
      if (u < 0.0 || u > 1.0) continue;
      if (v < 0.0 || u + v > 1.0) continue;
      if (t < 0.0 || t > h.t) continue;
      h.t = t;
      h.u = u;
      h.v = v;
      h.slot = leaf_first + i;
    }
    leaf_cnt = 0;
  }
  c.nodes += nnodes;
  c.tris += ntris;
  c.rays += 1;
}

// ---- the Intersection record of one traversed ray (BVHAccel::Traverse's out-parameter, bvh_accel.cc:773-844) -------------
// A miss leaves t = DBL_MAX, u = v = 0, faceID = -1 (bvh_accel.cc:782-786) and every other field zero; a NaN t (NaN ray) is
// reported as a miss by Traverse (`isect.t < DBL_MAX`, :838) although TestLeafNode already wrote faceID / materialID; a hit
// goes through BuildIntersection (bvh_accel.cc:699-769).  `is` may point into LDS, device or mapped host memory.
__device__ __forceinline__ void fill_intersection(const DScene &sc, V3 org, V3 dir, const Hit &h, bool hit, MgpuIntersection *is) {
  is->t = h.t; is->u = h.u; is->v = h.v;
  uint32_t faceID = 0xFFFFFFFFu, materialID = 0, f0 = 0, f1 = 0, f2 = 0;
  V3 pos = v3(0, 0, 0), gn = v3(0, 0, 0), sn = v3(0, 0, 0);
  double tc0 = 0.0, tc1 = 0.0;
  if (!hit && h.slot != kNoHit) {
    faceID = sc.tris[h.slot].face;
    materialID = sc.tris[h.slot].mat;
  }
  if (hit) {
    const DTri *tp = sc.tris + h.slot;
    const uint32_t face = tp->face;
    faceID = face;
    materialID = tp->mat;
    f0 = sc.faces[3 * (size_t)face + 0];
    f1 = sc.faces[3 * (size_t)face + 1];
    f2 = sc.faces[3 * (size_t)face + 2];
    pos = v3(org.x + h.t * dir.x, org.y + h.t * dir.y, org.z + h.t * dir.z);
    const V3 e1 = v3(tp->e1[0], tp->e1[1], tp->e1[2]), e2 = v3(tp->e2[0], tp->e2[1], tp->e2[2]);
    gn = normalized(cross(e1, e2));
    if (sc.fv_normals) {
      const double *nn = sc.fv_normals + 9 * (size_t)face;
      const double w = 1.0 - h.u - h.v;
      sn = v3(w * nn[0] + h.u * nn[3] + h.v * nn[6], w * nn[1] + h.u * nn[4] + h.v * nn[7],
              w * nn[2] + h.u * nn[5] + h.v * nn[8]);
    } else {
      sn = gn;
    }
    if (sc.fv_uvs) {
      const double *uv = sc.fv_uvs + 6 * (size_t)face;
      const double w = 1.0 - h.u - h.v;
      tc0 = w * uv[0] + h.u * uv[2] + h.v * uv[4];
      tc1 = w * uv[1] + h.u * uv[3] + h.v * uv[5];
    }
  }
  is->faceID = faceID; is->materialID = materialID; is->f0 = f0; is->f1 = f1; is->f2 = f2; is->pad_ = 0;
  is->position[0] = pos.x; is->position[1] = pos.y; is->position[2] = pos.z;
  is->geometricNormal[0] = gn.x; is->geometricNormal[1] = gn.y; is->geometricNormal[2] = gn.z;
  is->normal[0] = sn.x; is->normal[1] = sn.y; is->normal[2] = sn.z;
  for (int k = 0; k < 3; ++k) { is->tangent[k] = 0.0; is->binormal[k] = 0.0; }
  is->texcoord[0] = tc0; is->texcoord[1] = tc1;
}

// ---- Plane::intersect (prim-plane.cc:8-44): float core, double outputs ------------------------------------------------
// Returns true and overwrites t / normal when the plane is hit closer than `t`.
// `unit_n` = normalized((double)pl[0..2]) computed once on the host (the same IEEE sqrt / division).
__device__ __forceinline__ bool plane_hit(const float pl[4], const double unit_n[3], V3 org, V3 dir, double &t_io, V3 &normal) {
  V3 n = v3((double)pl[0], (double)pl[1], (double)pl[2]);
  const V3 v = normalized_w(dir);
  const float vn = (float)dot(v, n);
  if (fabsf(vn) > 1.1920929e-07f * 1024.0f) {
    const float on_d = (float)(dot(org, n) + (double)pl[3]);
    const float t = -on_d / vn;
    if ((t > 0) && ((double)t < t_io)) {
      t_io = (double)t;
      normal = v3(unit_n[0], unit_n[1], unit_n[2]);
      return true;
    }
  }
  return false;
}

// ---- GenerateBasis + SampleDiffuseIS (render.cc:271-339) ------------------------------------------------------------------
// TURN = false: the azimuth through the device library's sincospi; TURN = true: through sincos_turn and its table in LDS
// (mgpu_sincos.hpp), the same two numbers to 2e-16 for a quarter of the instructions (k_render_sm, k_render_env).
// A compile-time switch, not a null test of `azimuth`: the address of an LDS object may legitimately be 0, so the compiler kept
// both evaluations alive -- and hoisted the library polynomial's constants into six register pairs it then spilled (until
// round 4: 12 of the LDS variant's 13 spilled VGPRs served a branch that never ran).
template <bool TURN>
__device__ __forceinline__ V3 sample_diffuse_t(V3 n, Rng &rng, const SincosTable *azimuth) {
  // Minor axis by |n[i]| compared after rounding to float (fabsf); the loop of render.cc:279-285 keeps the FIRST
  // strict minimum below 1e6, and an index that stays -1 (NaN / huge normal) takes the final else branch.
  // Written as boolean selects, not as an int index + if/else-if chain: hipcc (ROCm 7.2, clang 22) lowers that
  // chain to a switch whose gfx950 code leaves the third case's tangent.x undefined (caught by the path probe
  // parity test; see DESIGN.md "Compiler hazards").
  const double ax = (double)fabsf((float)n.x), ay = (double)fabsf((float)n.y), az = (double)fabsf((float)n.z);
  const bool x_ok = ax < 1.0e+6;
  const double m0 = x_ok ? ax : 1.0e+6;
  const bool y_less = ay < m0;
  const double m1 = y_less ? ay : m0;
  const bool z_less = az < m1;
  const bool use_z = z_less || (!y_less && !x_ok); // index == 2 or index == -1: tangent = (-n.y, n.x, 0)
  const bool use_y = !z_less && y_less;            // index == 1:               tangent = (-n.z, 0, n.x)
  V3 t;                                            // index == 0:               tangent = (0, -n.z, n.y)
  t.x = use_z ? -n.y : (use_y ? -n.z : 0.0);
  t.y = use_z ? n.x : (use_y ? 0.0 : -n.z);
  t.z = use_z ? 0.0 : (use_y ? n.x : n.y);
  t = normalized_w(t);
  const V3 b = normalized_w(cross(t, n));
  // theta = acos(sqrt(1 - u1)), phi = 2*pi*u2 (render.cc:325-326); only sin/cos of the two angles are ever used.
  // = cos(theta) before the reference's acos -> cos round trip; the argument is 1 - k / 2^32 >= 2^-32
  const double x = sqrt_core(1.0 - rng_next(rng));
  const uint32_t k2 = rng_next_u32(rng);
  const double u2 = k2 * (1.0 / 4294967296.0);
  double sin_theta, cos_theta, sin_phi, cos_phi;
#if MGPU_SAMPLE_MATH == 0
  // literal transcription: device acos / sincos of the same arguments (<= 1 ulp from glibc's results)
  const double theta = acos(x);
  const double phi = 6.283185307179586 * u2; // 2.0 * M_PI, folded exactly
  sincos(theta, &sin_theta, &cos_theta);
  sincos(phi, &sin_phi, &cos_phi);
#else
  // same quantities without the inverse-function round trip: cos(acos(x)) = x and sin(acos(x)) = sqrt(1 - x^2)
  // (1 - x^2 with a single rounding), both within 1 ulp of the exact value the reference approximates to ~2 ulp;
  // sin/cos(2*pi*u2) through sincospi of the exactly representable 2*u2 (the reference rounds 2*pi*u2 first:
  // <= 1.5e-15 absolute difference).  See DESIGN.md "Numerics" for why this cannot move a pixel.
  cos_theta = x;
  const double s2 = fma(-x, x, 1.0); // >= 2^-53 or exactly 0 (u1 == 0)
  if (__builtin_expect(MGPU_ANY(!(s2 > 0x1p-100)), 0)) sin_theta = sqrt(s2);
  else sin_theta = sqrt_core(s2);
  if constexpr (TURN) sincos_turn(k2, *azimuth, sin_phi, cos_phi);
  else sincospi(2.0 * u2, &sin_phi, &cos_phi);
#endif
  const V3 T = scale(scale(t, cos_phi), sin_theta);
  const V3 B = scale(scale(b, sin_phi), sin_theta);
  const V3 N = scale(n, cos_theta);
  return (T + B) + N;
}

__device__ __forceinline__ V3 sample_diffuse(V3 n, Rng &rng) { return sample_diffuse_t<false>(n, rng, nullptr); }
__device__ __forceinline__ V3 sample_diffuse(V3 n, Rng &rng, const SincosTable *azimuth) { return sample_diffuse_t<true>(n, rng, azimuth); }

// Camera::GenerateRay direction (camera.cc:222-240)
__device__ __forceinline__ V3 camera_dir(const double *frame, double u, double v) {
  V3 d;
  d.x = (frame[3] + u * frame[6] + v * frame[9]) - frame[0];
  d.y = (frame[4] + u * frame[7] + v * frame[10]) - frame[1];
  d.z = (frame[5] + u * frame[8] + v * frame[11]) - frame[2];
  return normalized_w(d);
}

} // namespace mgpu
