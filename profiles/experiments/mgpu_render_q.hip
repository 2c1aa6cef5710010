// EXPERIMENT, NOT PART OF THE BUILD (kept with its measurements; see profiles/experiments/README.md and DESIGN.md 4.1).
// To try it again: copy next to mgpu_render_sm.hip, add it to SOURCES in mallie_amd/build.py, declare QParams /
// launch_render_q / render_q_slots in mgpu_kernels.hpp, add plane_t() (below) to mgpu_device.hpp and a kernel choice "q"
// in mgpu_render_strips_device that allocates blocks * ns * 56 bytes of path state.
//
// plane_t(): the ray-only half of Plane::intersect (prim-plane.cc:8-26), -1 when the plane cannot be hit:
//   n = (double)pl[0..2]; v = normalized(dir); vn = (float)dot(v, n); if (fabsf(vn) > 1024 * FLT_EPSILON) {
//   t = -(float)(dot(org, n) + (double)pl[3]) / vn; if (t > 0) return t; } return -1.0f;
//
// mgpu_render_q.hip -- k_render_q: the path tracer of k_render_sm with traversal and shading on DIFFERENT waves.
//
// In k_render_sm a lane that finishes its ray waits in the SHADE state until half the wave has finished too, and the
// SHADE step then runs its three sub-bodies (close a path / bounce / start a path) for the third of those lanes each
// concerns: shading costs about half of all issue slots at ~20 % useful lanes.  Here a workgroup of 16 waves (one per
// CU) keeps its rays in an LDS pool of `ns` slots and splits the work by wave:
//
//   traverser waves (12): lane states NODE / TRI as in k_render_sm, plus SWAP: hand the finished ray's hit back to its
//        slot, queue the slot for shading, take a READY slot and walk that ray -- no lane ever waits for shading;
//   shader waves (4):  take 64 slots at a time from the BOUNCE queue or from the END queue (the traverser sorted them:
//        a ray that hit something and may go on / a path that is over), so each batch runs ONE sub-body with all
//        lanes active, write the next ray into the slot and queue it READY.
//
// Slot = one path in flight: ray record in LDS (org, dir, 1/dir or the hit, plane distance / triangle slot, flags),
// path state (throughput, RNG, pixel, pass, length, material id) in HBM, indexed by slot.  Queues are rings of 16-bit
// slot ids in LDS: a producer reserves positions with one wave-level atomic and publishes by storing the id, a consumer
// claims positions with a CAS on the head and waits for the ids to appear.  Per-path arithmetic and operation order are
// those of k_render_sm (and PathTrace), so images and counters are identical; only who does what when changes.
#include "mgpu_device.hpp"
#include "mgpu_kernels.hpp"

namespace mgpu {

namespace {
enum : int { QT_NODE = 0, QT_TRI = 1, QT_SWAP = 2, QT_EXIT = 3 }; // SWAP: give a finished ray back and / or take a ready one
constexpr int kQBlock = 1024, kQWaves = 16;
constexpr uint32_t kRing = 2048, kRingMask = kRing - 1, kEmpty = 0xFFFFu;
// record flags (high word of field 9)
constexpr uint32_t kFlagCanBounce = 1u, kFlagPlaneOk = 2u, kFlagFresh = 4u;
} // namespace

#ifndef MGPU_Q_SHADERS
#define MGPU_Q_SHADERS 4
#endif
#ifndef MGPU_Q_SWAP_MIN
#define MGPU_Q_SWAP_MIN 8
#endif

struct QShared {
  RenderParams P;
  unsigned long long wg_cursor; // pixel items: hi32 = end, lo32 = next, (shard << 28) | index (see k_render_sm)
  unsigned long long cur;       // the item being handed out: hi32 = its code, lo32 = paths of it already taken (64 = none left)
  uint32_t wg_lock, wg_shard_off, wg_dry;
  uint32_t head[3], tail[3];    // queues: 0 = READY, 1 = BOUNCE, 2 = END
  uint32_t live;                // slots that still carry (or may get) a path
  unsigned long long cnt[5];
};

template <int CAP>
__global__ __launch_bounds__(kQBlock, 4) void k_render_q(DScene sc, RenderParams P_arg, QParams Q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ QShared S;
  constexpr int kTrav = kQWaves - MGPU_Q_SHADERS;
  const uint32_t ns = Q.ns;
  uint32_t *s_stack = reinterpret_cast<uint32_t *>(smem);                                   // [kTrav][CAP][64]
  unsigned long long *rec = reinterpret_cast<unsigned long long *>(smem + (size_t)kTrav * CAP * 256); // [10][ns]
  unsigned short *ring = reinterpret_cast<unsigned short *>(rec + (size_t)10 * ns);         // [3][kRing]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  if (threadIdx.x == 0) {
    S.P = P_arg;
    S.wg_cursor = 0ull; S.cur = 64ull; S.wg_lock = 0u; S.wg_shard_off = 0u; S.wg_dry = 0u;
    S.head[0] = S.head[1] = S.head[2] = 0u;
    S.tail[0] = S.tail[1] = 0u;
    S.tail[2] = ns; // every slot starts in the END queue, flagged FRESH: "give me my first path"
    S.live = ns;
    for (int k = 0; k < 5; ++k) S.cnt[k] = 0ull;
  }
  for (uint32_t i = threadIdx.x; i < 3 * kRing; i += kQBlock) ring[i] = (unsigned short)kEmpty;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < ns; i += kQBlock) {
    ring[2 * kRing + i] = (unsigned short)i;
    rec[(size_t)9 * ns + i] = (unsigned long long)kFlagFresh << 32;
  }
  __syncthreads();
  const RenderParams &P = S.P;

  // path state of this workgroup's slots in HBM (SoA)
  double *ps_thr = Q.ps_thr + (size_t)blockIdx.x * 3 * ns;         // [3][ns]
  uint4 *ps_rng = Q.ps_rng + (size_t)blockIdx.x * ns;              // [ns]
  uint32_t *ps_pix = Q.ps_u32 + (size_t)blockIdx.x * 3 * ns;       // [ns] lx | ly << 16
  uint32_t *ps_meta = ps_pix + ns;                                 // [ns] pass | pathLength << 16
  uint32_t *ps_mat = ps_pix + 2 * ns;                              // [ns] last material id
  float *ps_tp = Q.ps_tp + (size_t)blockIdx.x * ns;               // [ns] plane distance along the slot's ray (plane_t)

  auto rd = [&](int f, uint32_t s) -> double { return __longlong_as_double((long long)rec[(size_t)f * ns + s]); };
  auto wr = [&](int f, uint32_t s, double v) { rec[(size_t)f * ns + s] = (unsigned long long)__double_as_longlong(v); };

  // ---- queue primitives (wave-level; every lane of the wave calls them) ---------------------------------------------
  // push: lanes with `doit` append `slot` to ring q
  auto push = [&](int q, bool doit, uint32_t slot) {
    const unsigned long long m = __ballot(doit);
    if (!m) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&S.tail[q], (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (doit) {
      const uint32_t pos = (base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & kRingMask;
      unsigned short *e = &ring[q * kRing + pos];
      // the position's previous tenant (kRing pushes ago) has been claimed long since; wait until it has been read too
      while (__hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != kEmpty) __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(e, (unsigned short)slot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  // pop: up to `want` (<= 64) entries; lanes 0 .. n-1 receive one each; returns n.  need_full: all or nothing.
  auto pop = [&](int q, uint32_t want, bool need_full, uint32_t &slot) -> uint32_t {
    uint32_t n = 0, h = 0;
    if (lane == 0) {
      for (;;) {
        h = __hip_atomic_load(&S.head[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t t = __hip_atomic_load(&S.tail[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t avail = t - h;
        n = avail < want ? avail : want;
        if (n == 0 || (need_full && n < want)) { n = 0; break; }
        if (atomicCAS(&S.head[q], h, h + n) == h) break;
      }
    }
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
    if ((uint32_t)lane < n) {
      unsigned short *e = &ring[q * kRing + ((h + (uint32_t)lane) & kRingMask)];
      uint32_t v;
      while ((v = __hip_atomic_load(e, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == kEmpty) __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(e, (unsigned short)kEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      slot = v;
    }
    return n;
  };
  auto lds_load = [&](uint32_t *p) -> uint32_t {
    uint32_t v = 0;
    if (lane == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  };

  uint32_t c_trace_calls = 0, c_rays = 0, c_nodes = 0, c_tris = 0, c_paths = 0;

  if (wave < kTrav) {
    // =================================================== traverser ===================================================
    Stack<CAP, false> stk;
    stk.lds = s_stack + ((size_t)wave * CAP) * 64 + lane;
    stk.overflow = nullptr;
    int st = QT_SWAP;
    bool have_ray = false;
    uint32_t slot = 0, flags = 0;
    V3 org = v3(0, 0, 0), dir = v3(0, 0, 1);
    double ix = 0, iy = 0, iz = 0;
    bool sx = false, sy = false, sz = false;
    int sp = -1;
    double bt = kDblMax, bu = 0, bv = 0;
    uint32_t bslot = kNoHit;
    uint32_t tri_cur = 0, tri_end = 0;
    unsigned wd = 0; (void)wd;
#ifdef MGPU_UTIL
    unsigned long long u_node = 0, u_node_l = 0, u_tri = 0, u_tri_l = 0, u_swap = 0, u_swap_l = 0, u_starved = 0, u_idle = 0;
    unsigned long long t_node = 0, t_tri = 0, t_swap = 0, t_idle = 0, t0 = clock64(), t_all0 = clock64();
#endif
    for (;;) {
      const unsigned long long mN = __ballot(st == QT_NODE);
      const unsigned long long mT = __ballot(st == QT_TRI);
      const unsigned long long mW = __ballot(st == QT_SWAP);
      const int cN = __popcll(mN), cT = __popcll(mT), cW = __popcll(mW);
      if ((cN | cT | cW) == 0) break;
#ifdef MGPU_Q_WATCHDOG
      if (++wd > (unsigned)MGPU_Q_WATCHDOG) {
        if (lane == 0 && P_arg.stats) {
          atomicAdd(&P_arg.stats[16], 1ull);
          if (blockIdx.x == 0 && wave == 0) {
            P_arg.stats[17] = ((unsigned long long)cN << 32) | ((unsigned long long)cT << 16) | (unsigned long long)cW;
            P_arg.stats[18] = ((unsigned long long)S.head[0] << 32) | S.tail[0];
            P_arg.stats[19] = ((unsigned long long)S.head[1] << 32) | S.tail[1];
            P_arg.stats[20] = ((unsigned long long)S.head[2] << 32) | S.tail[2];
            P_arg.stats[21] = S.live;
          }
        }
        break;
      }
#endif
      // SWAP runs when enough lanes sit in it AND it can do something for them: some have a ray to give back, or rays
      // are ready to be taken (else the walkers would be starved by lanes that can only wait)
      bool run_swap = (cN == 0 && cT == 0);
      if (!run_swap && cW >= MGPU_Q_SWAP_MIN) {
        run_swap = __ballot(st == QT_SWAP && have_ray) != 0ull;
        if (!run_swap) {
          const uint32_t hR = lds_load(&S.head[0]);
          run_swap = lds_load(&S.tail[0]) != hR;
        }
      }
      if (!run_swap && cN >= cT) {
        // ================================ NODE step ================================
#ifdef MGPU_UTIL
        u_node++; u_node_l += cN; t0 = clock64();
#endif
        if (st == QT_NODE) {
#pragma unroll 1
          for (int rep = 0; rep < 4; ++rep) {
            const uint32_t ni = stk.get(sp);
            --sp;
            ++c_nodes;
            const MgpuNode *nd = sc.nodes + ni;
            const double2 b0 = *reinterpret_cast<const double2 *>(&nd->bmin[0]);
            const double2 b1 = *reinterpret_cast<const double2 *>(&nd->bmin[2]);
            const double2 b2 = *reinterpret_cast<const double2 *>(&nd->bmax[1]);
            const int4 meta = *reinterpret_cast<const int4 *>(&nd->flag);
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
                st = QT_TRI;
              }
            }
            if (st != QT_NODE || sp < 0) break;
          }
          if (st == QT_NODE && sp < 0) st = QT_SWAP;
        }
#ifdef MGPU_UTIL
        t_node += clock64() - t0;
#endif
      } else if (!run_swap) {
        // ================================ TRI step =================================
#ifdef MGPU_UTIL
        u_tri++; u_tri_l += cT; t0 = clock64();
#endif
        if (st == QT_TRI) {
#pragma unroll 1
          for (int rep = 0; rep < 16; ++rep) {
            const DTri *tp_ = sc.tris + tri_cur;
            const double2 a0 = reinterpret_cast<const double2 *>(tp_)[0];
            const double2 a1 = reinterpret_cast<const double2 *>(tp_)[1];
            const double2 a2 = reinterpret_cast<const double2 *>(tp_)[2];
            const double2 a3 = reinterpret_cast<const double2 *>(tp_)[3];
            const double e2z = tp_->e2[2];
            ++c_tris;
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
          if (tri_cur == tri_end) st = (sp < 0) ? QT_SWAP : QT_NODE;
        }
#ifdef MGPU_UTIL
        t_tri += clock64() - t0;
#endif
      } else {
        // ================================ SWAP step ================================
#ifdef MGPU_UTIL
        u_swap++; u_swap_l += cW; if (cN == 0 && cT == 0) u_idle++; t0 = clock64();
#endif
        const bool swap_lane = (st == QT_SWAP);
        // (1) hand the finished ray back: the hit replaces 1/dir in the record, the slot goes to the queue of the body
        //     that finishes it (the ray counts as a hit when a triangle was hit or the plane lies ahead of it)
        const bool giving = swap_lane && have_ray;
        bool to_bounce = false;
        if (giving) {
          wr(6, slot, bt); wr(7, slot, bu); wr(8, slot, bv);
          rec[(size_t)9 * ns + slot] = ((unsigned long long)flags << 32) | (unsigned long long)bslot;
          const bool hit = (bt < kDblMax) || (flags & kFlagPlaneOk);
          to_bounce = hit && (flags & kFlagCanBounce);
          have_ray = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        push(1, giving && to_bounce, slot);
        push(2, giving && !to_bounce, slot);
        // (2) take READY slots
        uint32_t got = 0;
        const uint32_t n = pop(0, (uint32_t)cW, false, got);
        // the n entries sit in lanes 0..n-1; give them to the first n swap lanes
        const uint32_t my_rank = (uint32_t)__popcll(mW & ((1ull << lane) - 1ull));
        const uint32_t src = __builtin_amdgcn_ds_bpermute((int)(my_rank << 2), (int)got);
        if (swap_lane && my_rank < n) {
          slot = src;
          org = v3(rd(0, slot), rd(1, slot), rd(2, slot));
          dir = v3(rd(3, slot), rd(4, slot), rd(5, slot));
          ix = rd(6, slot); iy = rd(7, slot); iz = rd(8, slot);
          flags = (uint32_t)(rec[(size_t)9 * ns + slot] >> 32);
          // BVHAccel::Traverse prologue, bvh_accel.cc:774-802
          sx = dir.x < 0.0; sy = dir.y < 0.0; sz = dir.z < 0.0;
          bt = kDblMax; bu = 0.0; bv = 0.0; bslot = kNoHit;
          sp = 0;
          stk.put(0, 0u);
          have_ray = true;
          ++c_rays;
          st = QT_NODE;
        }
#ifdef MGPU_UTIL
        if (n == 0) u_starved++;
#endif
        if (n == 0) {
          if (lds_load(&S.live) == 0) {
            if (swap_lane) st = QT_EXIT;
          } else if (cN == 0 && cT == 0) {
            __builtin_amdgcn_s_sleep(8); // nothing to walk and nothing ready yet
          }
        }
#ifdef MGPU_UTIL
        if (n == 0 && cN == 0 && cT == 0) t_idle += clock64() - t0; else t_swap += clock64() - t0;
#endif
      }
    }
#ifdef MGPU_UTIL
    if (lane == 0 && P_arg.stats) {
      atomicAdd(&P_arg.stats[8], u_node); atomicAdd(&P_arg.stats[9], u_node_l);
      atomicAdd(&P_arg.stats[10], u_tri); atomicAdd(&P_arg.stats[11], u_tri_l);
      atomicAdd(&P_arg.stats[12], u_swap); atomicAdd(&P_arg.stats[13], u_swap_l);
      atomicAdd(&P_arg.stats[14], u_starved); atomicAdd(&P_arg.stats[15], u_idle);
      atomicAdd(&P_arg.stats[27], t_node); atomicAdd(&P_arg.stats[28], t_tri); atomicAdd(&P_arg.stats[29], t_swap); atomicAdd(&P_arg.stats[30], t_idle); atomicAdd(&P_arg.stats[31], clock64() - t_all0);
    }
#endif
  } else {
    // ==================================================== shader =====================================================
    const int win_w = P.x1 - P.x0;
    const uint32_t tiles_x = (uint32_t)(win_w + 7) >> 3, tiles_y = (uint32_t)(P.n_rows + 7) >> 3;
    const uint32_t total_items = tiles_x * tiles_y * (uint32_t)P.passes;
    const uint32_t shard_items = (total_items + (uint32_t)kShards - 1) / (uint32_t)kShards;
    constexpr uint32_t kWgChunk = 8;
    uint32_t home_shard = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(home_shard));
    home_shard &= 7u;

    // writes the ray (org, dir) of path state (flags) into slot s and queues it READY
    auto emit_ray = [&](bool doit, uint32_t s, V3 o, V3 d, int pathLength) {
      if (doit) {
        wr(0, s, o.x); wr(1, s, o.y); wr(2, s, o.z);
        wr(3, s, d.x); wr(4, s, d.y); wr(5, s, d.z);
        wr(6, s, 1.0 / d.x); wr(7, s, 1.0 / d.y); wr(8, s, 1.0 / d.z); // no zero guard, as the reference
        const float tpl = P.has_plane ? plane_t(P.plane, o, d) : -1.0f;
        uint32_t f = (pathLength < P.maxPathLength) ? kFlagCanBounce : 0u;
        if (tpl > 0.0f && (double)tpl < kDblMax) f |= kFlagPlaneOk;
        rec[(size_t)9 * ns + s] = (unsigned long long)f << 32; // low word: the closest triangle's slot, on the way back
        ps_tp[s] = tpl;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      push(0, doit, s);
    };

    unsigned wd = 0; (void)wd;
#ifdef MGPU_UTIL
    unsigned long long u_b = 0, u_b_l = 0, u_e = 0, u_e_l = 0, u_sleep = 0, u_cyc_b = 0, u_cyc_e = 0, u_t0 = 0, u_samples = 0, u_sR = 0, u_sB = 0, u_sE = 0;
#endif
    for (;;) {
      if (lds_load(&S.live) == 0) break;
#ifdef MGPU_Q_WATCHDOG
      if (++wd > (unsigned)MGPU_Q_WATCHDOG) {
        if (lane == 0 && P_arg.stats) {
          atomicAdd(&P_arg.stats[22], 1ull);
          if (blockIdx.x == 0 && wave == kTrav) {
            P_arg.stats[23] = ((unsigned long long)S.head[0] << 32) | S.tail[0];
            P_arg.stats[24] = ((unsigned long long)S.head[1] << 32) | S.tail[1];
            P_arg.stats[25] = ((unsigned long long)S.head[2] << 32) | S.tail[2];
            P_arg.stats[26] = S.live;
            P_arg.stats[27] = S.cur;
          }
        }
        break;
      }
#endif
      // head first, tail second: heads only chase tails, so the differences cannot come out negative
      const uint32_t hB = lds_load(&S.head[1]), hE = lds_load(&S.head[2]), hR = lds_load(&S.head[0]);
      const uint32_t nB = lds_load(&S.tail[1]) - hB;
      const uint32_t nE = lds_load(&S.tail[2]) - hE;
      const uint32_t nR = lds_load(&S.tail[0]) - hR;
      const bool flush = nR < 128; // traversers are about to run dry: shade whatever has come back
#ifdef MGPU_UTIL
      u_samples++; u_sR += nR; u_sB += (nB > 4096 ? 0 : nB); u_sE += (nE > 4096 ? 0 : nE);
#endif
      int q = -1;
      if (nB >= 64 || nE >= 64) q = (nB >= nE) ? 1 : 2;
      else if (flush && (nB | nE) != 0) q = (nB >= nE) ? 1 : 2;
      if (q < 0) {
#ifdef MGPU_UTIL
        u_sleep++;
#endif
        __builtin_amdgcn_s_sleep(4);
        continue;
      }
      uint32_t s = 0;
      const uint32_t n = pop(q, 64, false, s);
      if (n == 0) continue;
      const bool act = (uint32_t)lane < n;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      // common: the finished ray and the path it belongs to
      V3 org = v3(0, 0, 0), dir = v3(0, 0, 1);
      double bt = kDblMax, bu = 0, bv = 0;
      uint32_t bslot = kNoHit, flags = kFlagFresh;
      float tp = -1.0f;
      double thr0 = 1, thr1 = 1, thr2 = 1;
      int pathLength = 1, pass = 0;
      uint32_t lx = 0, ly = 0, last_mat = kNoMaterial;
      Rng rng{1, 0, 0, 0};
      bool fresh = true;
      if (act) {
        const unsigned long long pk = rec[(size_t)9 * ns + s];
        flags = (uint32_t)(pk >> 32);
        fresh = (flags & kFlagFresh) != 0;
        if (!fresh) {
          bslot = (uint32_t)pk;
          org = v3(rd(0, s), rd(1, s), rd(2, s));
          dir = v3(rd(3, s), rd(4, s), rd(5, s));
          bt = rd(6, s); bu = rd(7, s); bv = rd(8, s);
          thr0 = ps_thr[s]; thr1 = ps_thr[ns + s]; thr2 = ps_thr[2 * ns + s];
          const uint32_t meta = ps_meta[s];
          pass = (int)(meta & 0xffffu);
          pathLength = (int)(meta >> 16);
          last_mat = ps_mat[s];
          const uint32_t pix = ps_pix[s];
          lx = pix & 0xffffu; ly = pix >> 16;
          tp = ps_tp[s];
        }
      }
      // what Scene::Trace + Plane::intersect leave behind (render.cc:403-408)
      bool hit = bt < kDblMax; // bvh_accel.cc:838
      double t = bt;
      bool plane_won = false;
      if (act && !fresh) {
        if (bslot != kNoHit) last_mat = sc.tris[bslot].mat; // written by TestLeafNode on every accepted triangle
        if (tp > 0.0f && (double)tp < t) { // Plane::intersect accepts (prim-plane.cc:27-37)
          t = (double)tp;
          hit = true;
          plane_won = true;
          last_mat = kNoMaterial; // prim-plane.cc:34
        }
      }
      const bool bounce = act && !fresh && hit && pathLength < P.maxPathLength;
      const bool ending = act && !bounce; // fresh slots "end" a path that never was: they just ask for a pixel
#ifdef MGPU_UTIL
      u_t0 = clock64();
#endif
      if (__ballot(bounce)) {
#ifdef MGPU_UTIL
        u_b++; u_b_l += __popcll(__ballot(bounce));
#endif
        // ---------------- BOUNCE: the rest of one PathTrace iteration (render.cc:414-452) ----------------
        V3 no = org, nd = dir;
        if (bounce) {
          V3 n;
          if (plane_won) {
            n = v3(P.plane_n[0], P.plane_n[1], P.plane_n[2]);
          } else if (sc.has_fv_normals) { // barycentric lerp, not renormalised (bvh_accel.cc:745-748)
            const double *nn = sc.slot_normal + 9 * (size_t)bslot;
            const double w = 1.0 - bu - bv;
            n = v3(w * nn[0] + bu * nn[3] + bv * nn[6], w * nn[1] + bu * nn[4] + bv * nn[7],
                   w * nn[2] + bu * nn[5] + bv * nn[8]);
          } else {
            const double *gn = sc.slot_normal + 3 * (size_t)bslot;
            n = v3(gn[0], gn[1], gn[2]);
          }
          const uint4 r4 = ps_rng[s];
          rng = Rng{r4.x, r4.y, r4.z, r4.w};
          const V3 hitP = org + scale(dir, t);
          (void)rng_next(rng); // `double r = randomreal();` drawn and never used (render.cc:430)
          const double ndoti = dot(n, neg(dir));
          if (ndoti < 0.0) n = neg(n);
          const V3 sd = sample_diffuse(n, rng);
          if (last_mat != kNoMaterial) { // Scene::GetMaterial, scene.h:58-65
            if ((size_t)(int)last_mat < (size_t)sc.nm) {
              thr0 *= sc.mat_diffuse[3 * (size_t)last_mat + 0];
              thr1 *= sc.mat_diffuse[3 * (size_t)last_mat + 1];
              thr2 *= sc.mat_diffuse[3 * (size_t)last_mat + 2];
            } else {
              thr0 *= 0.5; thr1 *= 0.5; thr2 *= 0.5;
            }
          }
          no = hitP + scale(sd, 1.0e-3);
          nd = sd;
          ++pathLength;
          ps_thr[s] = thr0; ps_thr[ns + s] = thr1; ps_thr[2 * ns + s] = thr2;
          ps_rng[s] = make_uint4(rng.x, rng.y, rng.z, rng.w);
          ps_meta[s] = (uint32_t)pass | ((uint32_t)pathLength << 16);
          ps_mat[s] = last_mat;
        }
        emit_ray(bounce, s, no, nd, pathLength);
      }
#ifdef MGPU_UTIL
      u_cyc_b += clock64() - u_t0; u_t0 = clock64();
#endif
      if (__ballot(ending)) {
#ifdef MGPU_UTIL
        u_e++; u_e_l += __popcll(__ballot(ending));
#endif
        // ---------------- END: close the path (render.cc:409-412, 419-421 + SURVEY F4), write the pixel ----------------
        if (ending && !fresh) {
          double rad0 = 0.0, rad1 = 0.0, rad2 = 0.0;
          if (!hit) {
            if (pathLength < 2) {
              c_trace_calls += 1; // eye ray -> background: radiance stays 0
            } else {
              // first miss of a path that has bounced: the reference iterates on to kMaxPathLength with the stale
              // record, every later ray ~1e308 away; same adds, same multiplies, same order, no ray
              c_trace_calls += (uint32_t)P.maxPathLength;
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
            c_trace_calls += (uint32_t)P.maxPathLength; // hit at the last allowed bounce
          }
          // image[...] = radiance (double -> float, render.cc:673-675); passes are summed later, in order
          float *dst = P.out + (size_t)pass * P.pass_stride + 3 * ((size_t)ly * (size_t)win_w + lx);
          dst[0] = (float)rad0;
          dst[1] = (float)rad1;
          dst[2] = (float)rad2;
        }
        // ---------------- hand-out: every ending slot wants its next path ----------------
        // The current item (8x8 tile, pass) and the position inside it are shared by the workgroup's shader waves
        // (S.cur: hi32 = item code, lo32 = paths already handed out): whichever wave closes paths continues where the
        // last one stopped, so no wave can be left sitting on the unused half of an item when the work runs out.
        bool want_pixel = ending, have_path = false;
        for (;;) {
          const unsigned long long want = __ballot(want_pixel);
          if (!want) break;
          uint32_t item_code = 0, pos = 0, n = 0, dry = 0;
          if (lane == 0) {
            const uint32_t k = (uint32_t)__popcll(want);
            for (;;) {
              const unsigned long long old = atomicAdd(&S.cur, 0ull);
              item_code = (uint32_t)(old >> 32);
              pos = (uint32_t)old;
              if (pos < 64u) {
                n = min(k, 64u - pos);
                if (atomicCAS(&S.cur, old, old + n) == old) break;
                continue;
              }
              // the item is used up: one wave fetches the next one (LDS cursor over the workgroup's reservation, refilled
              // from this XCD's global counter, then from the other XCDs'), the others retry
              if (__hip_atomic_load(&S.wg_dry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { dry = 1; break; }
              if (atomicCAS(&S.wg_lock, 0u, 1u) != 0u) {
                __builtin_amdgcn_s_sleep(2);
                continue;
              }
              if ((uint32_t)atomicAdd(&S.cur, 0ull) >= 64u) { // still used up (nobody advanced it while we took the lock)
                uint32_t next_code = 0;
                bool got = false;
                const unsigned long long c = atomicAdd(&S.wg_cursor, 0ull);
                if ((uint32_t)c < (uint32_t)(c >> 32)) {
                  next_code = (uint32_t)c;
                  atomicAdd(&S.wg_cursor, 1ull);
                  got = true;
                } else {
                  uint32_t off = S.wg_shard_off;
                  while (off < (uint32_t)kShards) {
                    const uint32_t sh = (home_shard + off) % (uint32_t)kShards;
                    const uint32_t base = atomicAdd(P.work_counter + sh, kWgChunk);
                    const uint32_t n_sh = sh * shard_items < total_items ? min(shard_items, total_items - sh * shard_items) : 0u;
                    if (base < n_sh) {
                      const uint32_t hi = (sh << 28) | min(base + kWgChunk, n_sh), lo = (sh << 28) | base;
                      next_code = lo;
                      atomicExch(&S.wg_cursor, ((unsigned long long)hi << 32) | (unsigned long long)(lo + 1u));
                      got = true;
                      break;
                    }
                    ++off;
                  }
                  S.wg_shard_off = off;
                }
                if (got) atomicExch(&S.cur, (unsigned long long)next_code << 32);
                else __hip_atomic_store(&S.wg_dry, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
              __threadfence_block();
              atomicExch(&S.wg_lock, 0u);
            }
          }
          dry = (uint32_t)__builtin_amdgcn_readfirstlane((int)dry);
          if (dry) break;
          item_code = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_code);
          pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
          n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
          const uint32_t item = (item_code >> 28) * shard_items + (item_code & 0x0fffffffu);
          const uint32_t item_tile = item / (uint32_t)P.passes;
          const uint32_t item_pass = item - item_tile * (uint32_t)P.passes;
          if (want_pixel) {
            const uint32_t rank = __popcll(want & ((1ull << lane) - 1ull));
            if (rank < n) {
              const uint32_t sl = pos + rank;
              const uint32_t x = (item_tile % tiles_x) * 8 + (sl & 7), y = (item_tile / tiles_x) * 8 + (sl >> 3);
              if (x < (uint32_t)win_w && y < (uint32_t)P.n_rows) { // slots of an edge tile outside the window are skipped
                lx = x; ly = y;
                pass = (int)item_pass;
                have_path = true;
                want_pixel = false;
              }
            }
          }
        }
        // ---------------- new eye path (PathTrace prologue, render.cc:387-400) or retire the slot ----------------
        V3 no = v3(0, 0, 0), nd = v3(0, 0, 1);
        if (ending && have_path) {
          const int gy = P.y_first + (int)(ly / (uint32_t)P.strip_h) * P.y_period + (int)(ly % (uint32_t)P.strip_h);
          const int gx = P.x0 + (int)lx;
          const uint32_t gpix = (uint32_t)gy * (uint32_t)P.W + (uint32_t)gx;
          uint32_t s4[4];
          if (P.rng_mode == MGPU_RNG_TABLE) {
            const uint4 q4 = reinterpret_cast<const uint4 *>(P.rng_states)[(size_t)pass * P.W * P.H + gpix];
            s4[0] = q4.x; s4[1] = q4.y; s4[2] = q4.z; s4[3] = q4.w;
          } else {
            hash_state(P.seed, P.pass_base + (uint32_t)pass, gpix, s4);
          }
          rng = Rng{s4[0], s4[1], s4[2], s4[3]};
          const float ju = (float)(rng_next(rng) - 0.5);
          const float jv = (float)(rng_next(rng) - 0.5);
          no = v3(P.frame[0], P.frame[1], P.frame[2]);
          nd = camera_dir(P.frame, (double)((float)gx + ju), (double)((float)gy + jv));
          ++c_paths;
          ps_thr[s] = 1.0; ps_thr[ns + s] = 1.0; ps_thr[2 * ns + s] = 1.0;
          ps_rng[s] = make_uint4(rng.x, rng.y, rng.z, rng.w);
          ps_pix[s] = lx | (ly << 16);
          ps_meta[s] = (uint32_t)pass | (1u << 16);
          ps_mat[s] = last_mat; // NOT reset: the reference's Intersection record is not cleared between paths either
        }
        emit_ray(ending && have_path, s, no, nd, 1);
        const uint32_t dead = (uint32_t)__popcll(__ballot(ending && !have_path));
        if (dead && lane == 0) atomicSub(&S.live, dead);
      }
#ifdef MGPU_UTIL
      u_cyc_e += clock64() - u_t0;
#endif
    }
#ifdef MGPU_UTIL
    if (lane == 0 && P_arg.stats) {
      atomicAdd(&P_arg.stats[16], u_b); atomicAdd(&P_arg.stats[17], u_b_l);
      atomicAdd(&P_arg.stats[18], u_e); atomicAdd(&P_arg.stats[19], u_e_l);
      atomicAdd(&P_arg.stats[20], u_sleep); atomicAdd(&P_arg.stats[21], u_cyc_b); atomicAdd(&P_arg.stats[22], u_cyc_e);
      atomicAdd(&P_arg.stats[23], u_samples); atomicAdd(&P_arg.stats[24], u_sR); atomicAdd(&P_arg.stats[25], u_sB); atomicAdd(&P_arg.stats[26], u_sE);
    }
#endif
  }

  // ---- counters: wave reduction, one LDS atomic per wave, one global atomic per workgroup and word ----
  unsigned long long v0 = c_trace_calls, v1 = c_rays, v2 = c_nodes, v3_ = c_tris, v4 = c_paths;
  for (int off = 32; off; off >>= 1) {
    v0 += __shfl_down(v0, off);
    v1 += __shfl_down(v1, off);
    v2 += __shfl_down(v2, off);
    v3_ += __shfl_down(v3_, off);
    v4 += __shfl_down(v4, off);
  }
  if (lane == 0) {
    atomicAdd(&S.cnt[0], v0);
    atomicAdd(&S.cnt[1], v1);
    atomicAdd(&S.cnt[2], v2);
    atomicAdd(&S.cnt[3], v3_);
    atomicAdd(&S.cnt[4], v4);
  }
  __syncthreads();
  if (threadIdx.x == 0 && P_arg.stats) {
    atomicAdd(&P_arg.stats[kStatTraceCalls], S.cnt[0]);
    atomicAdd(&P_arg.stats[kStatRays], S.cnt[1]);
    atomicAdd(&P_arg.stats[kStatNodes], S.cnt[2]);
    atomicAdd(&P_arg.stats[kStatTris], S.cnt[3]);
    atomicAdd(&P_arg.stats[kStatPaths], S.cnt[4]);
  }
}

size_t render_q_lds_bytes(int cap, uint32_t ns) {
  return (size_t)(kQWaves - MGPU_Q_SHADERS) * cap * 256 + (size_t)10 * ns * 8 + (size_t)3 * kRing * 2;
}

// largest slot count (multiple of 64, <= kRing) that fits next to the stacks; 0 when the pool would be smaller than
// the traverser lanes plus one batch
uint32_t render_q_slots(int cap) {
  const size_t fixed = (size_t)(kQWaves - MGPU_Q_SHADERS) * cap * 256 + (size_t)3 * kRing * 2 + sizeof(QShared) + 256;
  if (fixed >= kLdsBudget) return 0;
  size_t ns = (kLdsBudget - fixed) / 80;
  ns = ns / 64 * 64;
  if (ns > kRing) ns = kRing;
  const size_t lanes = (size_t)(kQWaves - MGPU_Q_SHADERS) * 64;
  return ns >= lanes + 128 ? (uint32_t)ns : 0u;
}

hipError_t launch_render_q(int cap, dim3 grid, hipStream_t s, const DScene &sc, const RenderParams &p, const QParams &q) {
  const size_t shmem = render_q_lds_bytes(cap, q.ns);
  static bool attr_done[2] = {false, false};
  if (cap == 16) {
    if (!attr_done[0]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_render_q<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
      if (e != hipSuccess) return e;
      attr_done[0] = true;
    }
    hipLaunchKernelGGL((k_render_q<16>), grid, dim3(kQBlock), shmem, s, sc, p, q);
  } else if (cap == 24) {
    if (!attr_done[1]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_render_q<24>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
      if (e != hipSuccess) return e;
      attr_done[1] = true;
    }
    hipLaunchKernelGGL((k_render_q<24>), grid, dim3(kQBlock), shmem, s, sc, p, q);
  } else {
    return hipErrorInvalidConfiguration;
  }
  return hipGetLastError();
}

} // namespace mgpu
