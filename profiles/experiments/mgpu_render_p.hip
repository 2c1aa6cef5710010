// EXPERIMENT, NOT PART OF THE BUILD (kept with its measurements; see profiles/experiments/README.md).
// To try it again: copy next to mgpu_render_sm.hip, add it to SOURCES in mallie_amd/build.py, declare kParkBytes
// (13 * 512 + 8 * 256) and launch_render_p in mgpu_kernels.hpp, and give mgpu_render_strips_device a kernel choice that
// allocates blocks * waves * 2 * kParkBytes bytes of parking space and calls launch_render_p instead of launch_render_sm.
//
// mgpu_render_p.hip -- k_render_p: k_render_sm with TWO paths per lane.
//
// k_render_sm: a lane whose ray is finished waits until half its wave has finished too, and the SHADE step then runs
// its sub-bodies (close a path + start one / bounce) for the part of those lanes each concerns.  Here every lane owns two
// path slots.  One ray is being walked (state in registers, as before); the other slot's ray is parked in memory -- waiting
// to be shaded, or shaded and armed.  A lane whose walk ends parks the hit and picks up its other slot's armed ray in a
// cheap SWAP step, so it keeps walking while the parked ray waits for ITS body: BOUNCE and END are separate steps with
// their own quorum, which they reach with 128 paths per wave instead of 64 and without idling anybody.
//
// Parked state lives in per-wave HBM rows of 64 entries (lane-indexed, so every access is one contiguous row):
//   record  10 x f64: org, dir, (1/dir | hit t, u, v), (flags << 32 | closest triangle slot)
//   path     3 x f64 throughput; 8 x u32: RNG state, pixel, pass | pathLength << 16, material id, plane distance
// Per-path arithmetic and operation order are those of k_render_sm (and PathTrace): images and counters are identical.
#include "mgpu_device.hpp"
#include "mgpu_kernels.hpp"

namespace mgpu {

namespace {
enum : int { PT_NODE = 0, PT_TRI = 1, PT_DONE = 2, PT_EMPTY = 3 };
enum : uint32_t { PS_NONE = 0, PS_TRAV = 1, PS_BOUNCE = 2, PS_END = 3, PS_READY = 4, PS_FRESH = 5 };
constexpr uint32_t kFlagCanBounce = 1u, kFlagPlaneOk = 2u;
} // namespace

#ifndef MGPU_P_BOUNCE_MIN
#define MGPU_P_BOUNCE_MIN 40
#endif
#ifndef MGPU_P_END_MIN
#define MGPU_P_END_MIN 40
#endif
#ifndef MGPU_P_SWAP_MIN
#define MGPU_P_SWAP_MIN 12
#endif

template <int CAP, bool LDS_SCENE, int BLOCK>
__global__ __launch_bounds__(BLOCK, 4) void k_render_p(DScene sc, RenderParams P_arg, unsigned char *park) {
  __shared__ RenderParams s_P;
  if (threadIdx.x == 0) s_P = P_arg;
  __syncthreads();
  const RenderParams &P = s_P;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int kWaves = BLOCK / 64;
  uint32_t *s_stack = reinterpret_cast<uint32_t *>(smem); // [kWaves][CAP][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Stack<CAP, false> stk;
  stk.lds = s_stack + ((size_t)wave * CAP) * 64 + lane;
  stk.overflow = nullptr;

  const unsigned char *lds_nodes = smem + (size_t)kWaves * CAP * 64 * sizeof(uint32_t);
  const unsigned char *lds_tris = lds_nodes + (size_t)P.lds_nodes_bytes;
  if (LDS_SCENE) {
    const uint4 *src = reinterpret_cast<const uint4 *>(sc.nodes);
    uint4 *dst = reinterpret_cast<uint4 *>(const_cast<unsigned char *>(lds_nodes));
    for (uint32_t i = threadIdx.x; i < (P.lds_nodes_bytes >> 4); i += BLOCK) dst[i] = src[i];
    const uint4 *src2 = reinterpret_cast<const uint4 *>(sc.tris);
    uint4 *dst2 = reinterpret_cast<uint4 *>(const_cast<unsigned char *>(lds_tris));
    for (uint32_t i = threadIdx.x; i < (P.lds_tris_bytes >> 4); i += BLOCK) dst2[i] = src2[i];
    __syncthreads();
  }

  const int win_w = P.x1 - P.x0;
  const uint32_t tiles_x = (uint32_t)(win_w + 7) >> 3, tiles_y = (uint32_t)(P.n_rows + 7) >> 3;
  const uint32_t total_items = tiles_x * tiles_y * (uint32_t)P.passes;
  uint32_t in_item = 64;
  bool exhausted = false;
  constexpr uint32_t kWgChunk = LDS_SCENE ? 16u : 8u;
  const uint32_t shard_items = (total_items + (uint32_t)kShards - 1) / (uint32_t)kShards;
  uint32_t home_shard = 0;
  uint32_t item_tile = 0, item_pass = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(home_shard));
  home_shard &= 7u;
  __shared__ unsigned long long wg_cursor;
  __shared__ uint32_t wg_lock, wg_shard_off, wg_dry;
  if (threadIdx.x == 0) {
    wg_cursor = 0ull;
    wg_lock = 0u;
    wg_shard_off = 0u;
    wg_dry = 0u;
  }
  __syncthreads();

  // ---- parked rows of this wave: [slot 0 | slot 1], lane-indexed --------------------------------------------------
  unsigned char *const pw = park + ((size_t)blockIdx.x * kWaves + wave) * 2 * kParkBytes;
  auto row_d = [&](uint32_t s, int f) -> double * { return reinterpret_cast<double *>(pw + (size_t)s * kParkBytes) + f * 64 + lane; };
  auto row_u = [&](uint32_t s, int r) -> uint32_t * {
    return reinterpret_cast<uint32_t *>(pw + (size_t)s * kParkBytes + 13 * 512) + r * 64 + lane;
  };

  // ---- the ray being walked (registers) ---------------------------------------------------------------------------
  int tst = PT_EMPTY;
  uint32_t tslot = 0, tflags = 0;
  V3 org = v3(0, 0, 0), dir = v3(0, 0, 1);
  double ix = 0, iy = 0, iz = 0;
  bool sx = false, sy = false, sz = false;
  int sp = -1;
  double bt = kDblMax, bu = 0, bv = 0;
  uint32_t bslot = kNoHit;
  uint32_t tri_cur = 0, tri_end = 0;
  // ---- the two slots' states ----------------------------------------------------------------------------------------
  uint32_t ps0 = PS_FRESH, ps1 = PS_FRESH;
  uint32_t n_rays = 0, n_nodes = 0, n_tris = 0, trace_calls = 0, paths = 0;

  // writes the armed ray (org, dir) of a path of length pathLength into slot s (the caller marks it READY)
  auto park_ray = [&](uint32_t s, V3 o, V3 d, int pathLength) {
    *row_d(s, 0) = o.x; *row_d(s, 1) = o.y; *row_d(s, 2) = o.z;
    *row_d(s, 3) = d.x; *row_d(s, 4) = d.y; *row_d(s, 5) = d.z;
    *row_d(s, 6) = 1.0 / d.x; *row_d(s, 7) = 1.0 / d.y; *row_d(s, 8) = 1.0 / d.z; // no zero guard, as the reference
    float tpl = -1.0f;
    if (P.has_plane) { // the ray-only half of Plane::intersect (prim-plane.cc:8-26)
      const V3 pn = v3((double)P.plane[0], (double)P.plane[1], (double)P.plane[2]);
      const V3 v = normalized(d);
      const float vn = (float)dot(v, pn);
      if (fabsf(vn) > 1.1920929e-07f * 1024.0f) {
        const float on_d = (float)(dot(o, pn) + (double)P.plane[3]);
        const float t = -on_d / vn;
        if (t > 0) tpl = t;
      }
    }
    uint32_t f = (pathLength < P.maxPathLength) ? kFlagCanBounce : 0u;
    if (tpl > 0.0f && (double)tpl < kDblMax) f |= kFlagPlaneOk;
    *reinterpret_cast<unsigned long long *>(row_d(s, 9)) = (unsigned long long)f << 32;
    *row_u(s, 7) = __float_as_uint(tpl);
  };

  for (;;) {
    const bool hasB = (ps0 == PS_BOUNCE) || (ps1 == PS_BOUNCE);
    const bool hasE = (ps0 == PS_END) || (ps1 == PS_END) || (ps0 == PS_FRESH) || (ps1 == PS_FRESH);
    const bool hasR = (ps0 == PS_READY) || (ps1 == PS_READY);
    const bool swapable = (tst == PT_DONE) || (tst == PT_EMPTY && hasR);
    const int cN = __popcll(__ballot(tst == PT_NODE)), cT = __popcll(__ballot(tst == PT_TRI));
    const int cB = __popcll(__ballot(hasB)), cE = __popcll(__ballot(hasE)), cW = __popcll(__ballot(swapable));
    if ((cN | cT | cB | cE | cW) == 0) break;
    const bool no_trav = (cN == 0 && cT == 0);
    int step; // 0 NODE, 1 TRI, 2 SWAP, 3 BOUNCE, 4 END
    if (cB >= MGPU_P_BOUNCE_MIN) step = 3;
    else if (cE >= MGPU_P_END_MIN) step = 4;
    else if (cW >= MGPU_P_SWAP_MIN || (no_trav && cW > 0)) step = 2;
    else if (!no_trav) step = (cN >= cT) ? 0 : 1;
    else step = (cB >= cE) ? 3 : 4;

    if (step == 0) {
      // ================================ NODE step ================================
      if (tst == PT_NODE) {
#pragma unroll 1
        for (int rep = 0; rep < 4; ++rep) {
          const uint32_t ni = stk.get(sp);
          --sp;
          ++n_nodes;
          double2 b0, b1, b2;
          int4 meta;
          if (LDS_SCENE) {
            const unsigned char *nd = lds_nodes + (size_t)ni * 64;
            b0 = *reinterpret_cast<const double2 *>(nd);
            b1 = *reinterpret_cast<const double2 *>(nd + 16);
            b2 = *reinterpret_cast<const double2 *>(nd + 32);
            meta = *reinterpret_cast<const int4 *>(nd + 48);
          } else {
            const MgpuNode *nd = sc.nodes + ni;
            b0 = *reinterpret_cast<const double2 *>(&nd->bmin[0]);
            b1 = *reinterpret_cast<const double2 *>(&nd->bmin[2]);
            b2 = *reinterpret_cast<const double2 *>(&nd->bmax[1]);
            meta = *reinterpret_cast<const int4 *>(&nd->flag);
          }
          // IntersectRayAABB, bvh_accel.cc:550-593
          const double nx = sx ? b1.y : b0.x, fx = sx ? b0.x : b1.y;
          const double ny = sy ? b2.x : b0.y, fy = sy ? b0.y : b2.x;
          const double nz = sz ? b2.y : b1.x, fz = sz ? b1.x : b2.y;
          const double tmin_x = (nx - org.x) * ix, tmax_x = (fx - org.x) * ix;
          const double tmin_y = (ny - org.y) * iy, tmax_y = (fy - org.y) * iy;
          double tmin = (tmin_x > tmin_y) ? tmin_x : tmin_y;
          double tmax = (tmax_x < tmax_y) ? tmax_x : tmax_y;
          const double tmin_z = (nz - org.z) * iz, tmax_z = (fz - org.z) * iz;
          tmin = (tmin > tmin_z) ? tmin : tmin_z;
          tmax = (tmax < tmax_z) ? tmax : tmax_z;
          const bool hit = (tmax > 0.0) && (tmin <= tmax) && (tmin <= bt);
          if (hit) {
            if (meta.x == 0) {
              const bool nearIsSecond = (meta.y == 0) ? sx : ((meta.y == 1) ? sy : sz); // dirSign[node.axis]
              const uint32_t c0 = (uint32_t)meta.z, c1 = (uint32_t)meta.w;
              stk.put(sp + 1, nearIsSecond ? c0 : c1); // far
              stk.put(sp + 2, nearIsSecond ? c1 : c0); // near: popped first
              sp += 2;
            } else if (meta.z != 0) {
              tri_cur = (uint32_t)meta.w;
              tri_end = (uint32_t)meta.w + (uint32_t)meta.z;
              tst = PT_TRI;
            }
          }
          if (tst != PT_NODE || sp < 0) break;
        }
        if (tst == PT_NODE && sp < 0) tst = PT_DONE;
      }
    } else if (step == 1) {
      // ================================ TRI step =================================
      if (tst == PT_TRI) {
#pragma unroll 1
        for (int rep = 0; rep < 16; ++rep) {
          double2 a0, a1, a2, a3;
          double e2z;
          if (LDS_SCENE) {
            const unsigned char *tp = lds_tris + (size_t)tri_cur * 80;
            a0 = *reinterpret_cast<const double2 *>(tp);
            a1 = *reinterpret_cast<const double2 *>(tp + 16);
            a2 = *reinterpret_cast<const double2 *>(tp + 32);
            a3 = *reinterpret_cast<const double2 *>(tp + 48);
            e2z = *reinterpret_cast<const double *>(tp + 64);
          } else {
            const DTri *tp = sc.tris + tri_cur;
            a0 = reinterpret_cast<const double2 *>(tp)[0];
            a1 = reinterpret_cast<const double2 *>(tp)[1];
            a2 = reinterpret_cast<const double2 *>(tp)[2];
            a3 = reinterpret_cast<const double2 *>(tp)[3];
            e2z = tp->e2[2];
          }
          ++n_tris;
          // TriangleIsect, bvh_accel.cc:595-638
          const V3 p0 = v3(a0.x, a0.y, a1.x), e1 = v3(a1.y, a2.x, a2.y), e2 = v3(a3.x, a3.y, e2z);
          const V3 p = cross(dir, e2);
          const double det = dot(e1, p);
          if (!(fabs(det) < kDblEps1024)) {
            const double invDet = 1.0 / det;
            const V3 s = org - p0;
            const V3 q = cross(s, e1);
            const double u = dot(s, p) * invDet;
            const double v = dot(q, dir) * invDet;
            const double t = dot(e2, q) * invDet;
            const bool rej = (u < 0.0 || u > 1.0) || (v < 0.0 || u + v > 1.0) || (t < 0.0 || t > bt);
            if (!rej) {
              bt = t;
              bu = u;
              bv = v;
              bslot = tri_cur;
            }
          }
          ++tri_cur;
          if (tri_cur == tri_end) break;
        }
        if (tri_cur == tri_end) tst = (sp < 0) ? PT_DONE : PT_NODE;
      }
    } else if (step == 2) {
      // ================================ SWAP step ================================
      if (tst == PT_DONE) { // park the hit in the walked ray's slot and queue it for the body that finishes it
        *row_d(tslot, 6) = bt; *row_d(tslot, 7) = bu; *row_d(tslot, 8) = bv;
        *reinterpret_cast<unsigned long long *>(row_d(tslot, 9)) = ((unsigned long long)tflags << 32) | (unsigned long long)bslot;
        const bool hit = (bt < kDblMax) || (tflags & kFlagPlaneOk);
        const uint32_t kind = (hit && (tflags & kFlagCanBounce)) ? PS_BOUNCE : PS_END;
        if (tslot == 0) ps0 = kind; else ps1 = kind;
        tst = PT_EMPTY;
      }
      if (tst == PT_EMPTY && (ps0 == PS_READY || ps1 == PS_READY)) { // pick the other slot's armed ray up
        const uint32_t s = (ps0 == PS_READY) ? 0u : 1u;
        org = v3(*row_d(s, 0), *row_d(s, 1), *row_d(s, 2));
        dir = v3(*row_d(s, 3), *row_d(s, 4), *row_d(s, 5));
        ix = *row_d(s, 6); iy = *row_d(s, 7); iz = *row_d(s, 8);
        tflags = (uint32_t)(*reinterpret_cast<unsigned long long *>(row_d(s, 9)) >> 32);
        if (s == 0) ps0 = PS_TRAV; else ps1 = PS_TRAV;
        tslot = s;
        // BVHAccel::Traverse prologue, bvh_accel.cc:774-802
        sx = dir.x < 0.0; sy = dir.y < 0.0; sz = dir.z < 0.0;
        bt = kDblMax; bu = 0.0; bv = 0.0; bslot = kNoHit;
        sp = 0;
        stk.put(0, 0u);
        ++n_rays;
        tst = PT_NODE;
      }
    } else if (step == 3) {
      // ================================ BOUNCE step ==============================
      // the rest of one PathTrace iteration (render.cc:403-452) for a parked ray that hit something and may go on
      if (hasB) {
        const uint32_t s = (ps0 == PS_BOUNCE) ? 0u : 1u;
        const V3 o = v3(*row_d(s, 0), *row_d(s, 1), *row_d(s, 2));
        const V3 d = v3(*row_d(s, 3), *row_d(s, 4), *row_d(s, 5));
        const double ht = *row_d(s, 6), hu = *row_d(s, 7), hv = *row_d(s, 8);
        const uint32_t hslot = (uint32_t)*reinterpret_cast<unsigned long long *>(row_d(s, 9));
        const float tp = __uint_as_float(*row_u(s, 7));
        const uint32_t meta = *row_u(s, 5);
        int pathLength = (int)(meta >> 16);
        uint32_t last_mat = *row_u(s, 6);
        bool hit = ht < kDblMax; // bvh_accel.cc:838
        double t = ht;
        bool plane_won = false;
        if (hslot != kNoHit) last_mat = sc.tris[hslot].mat; // written by TestLeafNode on every accepted triangle
        if (tp > 0.0f && (double)tp < t) { // Plane::intersect accepts (prim-plane.cc:27-37)
          t = (double)tp;
          hit = true;
          plane_won = true;
          last_mat = kNoMaterial; // prim-plane.cc:34
        }
        if (!hit || pathLength >= P.maxPathLength) {
          if (s == 0) ps0 = PS_END; else ps1 = PS_END; // only with non-finite distances: END redoes the ray
        } else {
          V3 n;
          if (plane_won) {
            n = v3(P.plane_n[0], P.plane_n[1], P.plane_n[2]);
          } else if (sc.has_fv_normals) { // barycentric lerp, not renormalised (bvh_accel.cc:745-748)
            const double *nn = sc.slot_normal + 9 * (size_t)hslot;
            const double w = 1.0 - hu - hv;
            n = v3(w * nn[0] + hu * nn[3] + hv * nn[6], w * nn[1] + hu * nn[4] + hv * nn[7],
                   w * nn[2] + hu * nn[5] + hv * nn[8]);
          } else {
            const double *gn = sc.slot_normal + 3 * (size_t)hslot;
            n = v3(gn[0], gn[1], gn[2]);
          }
          Rng rng{*row_u(s, 0), *row_u(s, 1), *row_u(s, 2), *row_u(s, 3)};
          const V3 hitP = o + scale(d, t);
          (void)rng_next(rng); // `double r = randomreal();` drawn and never used (render.cc:430)
          const double ndoti = dot(n, neg(d));
          if (ndoti < 0.0) n = neg(n);
          const V3 sd = sample_diffuse(n, rng);
          if (last_mat != kNoMaterial) { // Scene::GetMaterial, scene.h:58-65
            double m0 = 0.5, m1 = 0.5, m2 = 0.5;
            if ((size_t)(int)last_mat < (size_t)sc.nm) {
              m0 = sc.mat_diffuse[3 * (size_t)last_mat + 0];
              m1 = sc.mat_diffuse[3 * (size_t)last_mat + 1];
              m2 = sc.mat_diffuse[3 * (size_t)last_mat + 2];
            }
            *row_d(s, 10) *= m0; *row_d(s, 11) *= m1; *row_d(s, 12) *= m2;
          }
          ++pathLength;
          *row_u(s, 0) = rng.x; *row_u(s, 1) = rng.y; *row_u(s, 2) = rng.z; *row_u(s, 3) = rng.w;
          *row_u(s, 5) = (meta & 0xffffu) | ((uint32_t)pathLength << 16);
          *row_u(s, 6) = last_mat;
          park_ray(s, hitP + scale(sd, 1.0e-3), sd, pathLength);
          if (s == 0) ps0 = PS_READY; else ps1 = PS_READY;
        }
      }
    } else {
      // ================================ END step =================================
      // close a path (miss, or hit at the last allowed bounce), write its pixel, start the slot's next path
      const bool end_lane = hasE;
      const uint32_t s = ((ps0 == PS_END) || (ps0 == PS_FRESH)) ? 0u : 1u;
      bool want_pixel = false, have_path = false;
      uint32_t lx = 0, ly = 0, last_mat = kNoMaterial;
      int pass = 0;
      if (end_lane) {
        const bool fresh = ((s == 0) ? ps0 : ps1) == PS_FRESH;
        want_pixel = true;
        if (!fresh) {
          const double ht = *row_d(s, 6);
          const uint32_t hslot = (uint32_t)*reinterpret_cast<unsigned long long *>(row_d(s, 9));
          const float tp = __uint_as_float(*row_u(s, 7));
          const uint32_t meta = *row_u(s, 5);
          const int pathLength = (int)(meta >> 16);
          pass = (int)(meta & 0xffffu);
          last_mat = *row_u(s, 6);
          const uint32_t pix = *row_u(s, 4);
          lx = pix & 0xffffu; ly = pix >> 16;
          bool hit = ht < kDblMax;
          if (hslot != kNoHit) last_mat = sc.tris[hslot].mat;
          if (tp > 0.0f && (double)tp < ht) {
            hit = true;
            last_mat = kNoMaterial;
          }
          if (hit && pathLength < P.maxPathLength) {
            if (s == 0) ps0 = PS_BOUNCE; else ps1 = PS_BOUNCE; // only with non-finite distances
            want_pixel = false;
          } else {
            double rad0 = 0.0, rad1 = 0.0, rad2 = 0.0;
            if (!hit) {
              if (pathLength < 2) {
                trace_calls += 1; // eye ray -> background: radiance stays 0 (render.cc:409-412)
              } else {
                // first miss of a path that has bounced: the reference iterates on to kMaxPathLength with the stale
                // record, every later ray ~1e308 away; same adds, same multiplies, same order, no ray
                trace_calls += (uint32_t)P.maxPathLength;
                double thr0 = *row_d(s, 10), thr1 = *row_d(s, 11), thr2 = *row_d(s, 12);
                double d0 = 0.5, d1 = 0.5, d2 = 0.5; // Material().diffuse default (material.h:12-15)
                const bool mul = last_mat != kNoMaterial;
                if (mul && (size_t)(int)last_mat < (size_t)sc.nm) {
                  d0 = sc.mat_diffuse[3 * (size_t)last_mat + 0];
                  d1 = sc.mat_diffuse[3 * (size_t)last_mat + 1];
                  d2 = sc.mat_diffuse[3 * (size_t)last_mat + 2];
                }
                for (int L = pathLength;; ++L) {
                  const double dl = (double)(unsigned)L;
                  rad0 += thr0 * 0.5 / dl;
                  rad1 += thr1 * 0.5 / dl;
                  rad2 += thr2 * 0.5 / dl;
                  if (L >= P.maxPathLength) break;
                  if (mul) { thr0 *= d0; thr1 *= d1; thr2 *= d2; }
                }
              }
            } else {
              trace_calls += (uint32_t)P.maxPathLength; // hit at the last allowed bounce
            }
            // image[...] = radiance (double -> float, render.cc:673-675); passes are summed later, in order
            float *dst = P.out + (size_t)pass * P.pass_stride + 3 * ((size_t)ly * (size_t)win_w + lx);
            dst[0] = (float)rad0;
            dst[1] = (float)rad1;
            dst[2] = (float)rad2;
          }
        }
      }
      // ---- path hand-out, executed by the whole wave (the cursor variables are wave-uniform; as in k_render_sm) ----
      for (;;) {
        const unsigned long long want = __ballot(want_pixel);
        if (!want || exhausted) break;
        if (in_item >= 64) {
          uint32_t cur_shard = 0, item_local = 0;
          for (;;) {
            unsigned long long c = 0;
            if (lane == 0) c = atomicAdd(&wg_cursor, 1ull);
            const uint32_t nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)c);
            const uint32_t end = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(c >> 32));
            if (nxt < end) {
              cur_shard = nxt >> 28;
              item_local = nxt & 0x0fffffffu;
              break;
            }
            uint32_t flag = 0;
            if (lane == 0) flag = __hip_atomic_load(&wg_dry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__builtin_amdgcn_readfirstlane((int)flag)) { exhausted = true; break; }
            uint32_t won = 0;
            if (lane == 0) won = (atomicCAS(&wg_lock, 0u, 1u) == 0u) ? 1u : 0u;
            if (!__builtin_amdgcn_readfirstlane((int)won)) {
              __builtin_amdgcn_s_sleep(4);
              continue;
            }
            if (lane == 0) {
              const unsigned long long now = atomicAdd(&wg_cursor, 0ull);
              if ((uint32_t)now >= (uint32_t)(now >> 32)) {
                bool got = false;
                uint32_t off = wg_shard_off;
                while (off < (uint32_t)kShards) {
                  const uint32_t sh = (home_shard + off) % (uint32_t)kShards;
                  const uint32_t base = atomicAdd(P.work_counter + sh, kWgChunk);
                  const uint32_t n_sh = LDS_SCENE ? (total_items > sh ? (total_items - sh + (uint32_t)kShards - 1) / (uint32_t)kShards : 0u)
                                                  : (sh * shard_items < total_items ? min(shard_items, total_items - sh * shard_items) : 0u);
                  if (base < n_sh) {
                    const uint32_t hi = (sh << 28) | min(base + kWgChunk, n_sh), lo = (sh << 28) | base;
                    atomicExch(&wg_cursor, ((unsigned long long)hi << 32) | (unsigned long long)lo);
                    got = true;
                    break;
                  }
                  ++off;
                }
                wg_shard_off = off;
                if (!got) __hip_atomic_store(&wg_dry, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
              __threadfence_block();
              atomicExch(&wg_lock, 0u);
            }
          }
          if (exhausted) break;
          in_item = 0;
          const uint32_t item = LDS_SCENE ? item_local * (uint32_t)kShards + cur_shard : cur_shard * shard_items + item_local;
          const uint32_t ti = item / (uint32_t)P.passes;
          item_pass = item - ti * (uint32_t)P.passes;
          item_tile = P.tile_order ? (uint32_t)__builtin_amdgcn_readfirstlane((int)P.tile_order[ti]) : ti;
        }
        if (want_pixel) {
          const uint32_t rank = __popcll(want & ((1ull << lane) - 1ull));
          const uint32_t sl = in_item + rank;
          if (sl < 64) {
            const uint32_t x = (item_tile % tiles_x) * 8 + (sl & 7), y = (item_tile / tiles_x) * 8 + (sl >> 3);
            if (x < (uint32_t)win_w && y < (uint32_t)P.n_rows) { // slots of an edge tile outside the window are skipped
              lx = x; ly = y;
              pass = (int)item_pass;
              have_path = true;
              want_pixel = false;
            }
          }
        }
        in_item += (uint32_t)__popcll(want);
        if (in_item >= 64) in_item = 64;
      }
      // ---- the slot's next eye path (PathTrace prologue, render.cc:387-400) ----
      if (end_lane) {
        if (have_path) {
          const int gy = P.y_first + (int)(ly / (uint32_t)P.strip_h) * P.y_period + (int)(ly % (uint32_t)P.strip_h);
          const int gx = P.x0 + (int)lx;
          const uint32_t gpix = (uint32_t)gy * (uint32_t)P.W + (uint32_t)gx;
          uint32_t s4[4];
          if (P.rng_mode == MGPU_RNG_TABLE) {
            const uint4 q = reinterpret_cast<const uint4 *>(P.rng_states)[(size_t)pass * P.W * P.H + gpix];
            s4[0] = q.x; s4[1] = q.y; s4[2] = q.z; s4[3] = q.w;
          } else {
            hash_state(P.seed, P.pass_base + (uint32_t)pass, gpix, s4);
          }
          Rng rng{s4[0], s4[1], s4[2], s4[3]};
          const float ju = (float)(rng_next(rng) - 0.5);
          const float jv = (float)(rng_next(rng) - 0.5);
          const V3 o = v3(P.frame[0], P.frame[1], P.frame[2]);
          const V3 d = camera_dir(P.frame, (double)((float)gx + ju), (double)((float)gy + jv));
          ++paths;
          *row_d(s, 10) = 1.0; *row_d(s, 11) = 1.0; *row_d(s, 12) = 1.0;
          *row_u(s, 0) = rng.x; *row_u(s, 1) = rng.y; *row_u(s, 2) = rng.z; *row_u(s, 3) = rng.w;
          *row_u(s, 4) = lx | (ly << 16);
          *row_u(s, 5) = (uint32_t)pass | (1u << 16);
          *row_u(s, 6) = last_mat; // NOT reset: the reference's Intersection record is not cleared between paths either
          park_ray(s, o, d, 1);
          if (s == 0) ps0 = PS_READY; else ps1 = PS_READY;
        } else if (want_pixel) {
          if (s == 0) ps0 = PS_NONE; else ps1 = PS_NONE; // no pixel left: the slot retires
        }
      }
    }
  }

  // ---- counters: one atomic per wave and word -----------------------------------------------------------------
  unsigned long long v0 = trace_calls, v1 = n_rays, v2 = n_nodes, v3_ = n_tris, v4 = paths;
  for (int off = 32; off; off >>= 1) {
    v0 += __shfl_down(v0, off);
    v1 += __shfl_down(v1, off);
    v2 += __shfl_down(v2, off);
    v3_ += __shfl_down(v3_, off);
    v4 += __shfl_down(v4, off);
  }
  if (lane == 0 && P.stats) {
    atomicAdd(&P.stats[kStatTraceCalls], v0);
    atomicAdd(&P.stats[kStatRays], v1);
    atomicAdd(&P.stats[kStatNodes], v2);
    atomicAdd(&P.stats[kStatTris], v3_);
    atomicAdd(&P.stats[kStatPaths], v4);
  }
}

template <int CAP, bool LDS_SCENE, int BLOCK>
static hipError_t launch_one(dim3 grid, hipStream_t s, size_t shmem, const DScene &sc, const RenderParams &p, unsigned char *park) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_render_p<CAP, LDS_SCENE, BLOCK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL((k_render_p<CAP, LDS_SCENE, BLOCK>), grid, dim3(BLOCK), shmem, s, sc, p, park);
  return hipGetLastError();
}

hipError_t launch_render_p(int cap, bool lds_scene, dim3 grid, hipStream_t s, size_t shmem, const DScene &sc,
                           const RenderParams &p, unsigned char *park) {
  if (lds_scene) {
    if (cap == 16) return launch_one<16, true, 1024>(grid, s, shmem, sc, p, park);
    if (cap == 24) return launch_one<24, true, 1024>(grid, s, shmem, sc, p, park);
  } else {
    if (cap == 16) return launch_one<16, false, 256>(grid, s, shmem, sc, p, park);
    if (cap == 24) return launch_one<24, false, 256>(grid, s, shmem, sc, p, park);
    if (cap == 32) return launch_one<32, false, 256>(grid, s, shmem, sc, p, park);
  }
  return hipErrorInvalidConfiguration;
}

} // namespace mgpu
