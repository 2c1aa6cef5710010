// mgpu_trace_sm.hip -- k_trace_sm: batched Scene::Trace (scene.cc:253-315 -> BVHAccel::Traverse bvh_accel.cc:773-844 +
// BuildIntersection :699-769) with the wave-scheduled traversal of k_render_sm.
//
// k_trace (mgpu_kernels.hip) walks one ray per lane to completion; on incoherent rays in a large BVH a wave then waits
// for its slowest lane's chain of dependent HBM loads while 63 lanes idle.  Here the waves are persistent and every lane
// carries an explicit state
//
//     NODE : enter up to 4 interior nodes: both child boxes from one 128-byte record, near child entered directly,
//            far child stacked with its tmin (wide_node_step, mgpu_device.hpp)  (bvh_accel.cc:805-834, 550-593)
//     TRI  : test the open leaf's triangles in leaf order                      (bvh_accel.cc:595-697)
//     EMIT : write the finished ray's Intersection record and hit flag, take the next ray index from the wave's cursor,
//            load the ray and arm its traversal                                (bvh_accel.cc:774-802, 699-769, 838)
//
// and each trip of the wave loop runs the one body most lanes wait for.  A lane that finishes early emits and re-arms
// while its neighbours are still walking.  Per-ray operation order is that of BVHAccel::Traverse, so records, hit
// flags and the node / triangle visit counters are identical to k_trace's and the CPU's (asserted by the tests).
//
// Ray indices are handed out in order: a wave reserves kRayChunk consecutive indices with one global atomic and deals
// them to its lanes as they free up, so neighbouring lanes mostly hold neighbouring rays (coalesced-ish loads, record
// stores that fill whole cache lines between them).
#include "mgpu_device.hpp"
#include "mgpu_kernels.hpp"

#ifndef MGPU_TRACE_NODE_WEIGHT
#define MGPU_TRACE_NODE_WEIGHT 1 // NODE runs when cN * weight >= cT (camera rays: 1: 0.77, 2: 0.82, 4: 0.82 ms per 4M; incoherent: no difference)
#endif

namespace mgpu {

namespace {
enum : int { TS_NODE = 0, TS_TRI = 1, TS_EMIT = 2, TS_IDLE = 3 };
constexpr uint32_t kRayChunk = 256;
constexpr int kTraceBlock = 256;
} // namespace

#ifndef MGPU_EMIT_MIN
#define MGPU_EMIT_MIN 24
#endif

__global__ __launch_bounds__(kTraceBlock, 4) void k_trace_sm(DScene sc, const MgpuRay *__restrict__ rays, uint32_t n,
                                                            MgpuIntersection *__restrict__ out,
                                                            uint8_t *__restrict__ hit_out, uint32_t *work_counter,
                                                            unsigned long long *__restrict__ stats,
                                                            const uint32_t *__restrict__ select) {
  if (select && *select != kTraceSelectSm) return; // k_trace_probe chose the other kernel for this batch
  MGPU_DYN_SHARED(unsigned char, smem);
  __shared__ unsigned long long s_cnt[3];
  __shared__ unsigned char s_owner[kTraceBlock]; // TRI step with shared leaves: lane of the k-th open leaf, per wave
  using WS = WStack<kWideStackLds>;
  // per wave: the far-child stack's LDS part, then (all waves) the record staging: 16 records of 23 8-byte pieces + their
  // 16 ray indices per wave
  unsigned long long *s_stage = reinterpret_cast<unsigned long long *>(smem + (size_t)(kTraceBlock / 64) * WS::kWaveBytes);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t slot = (size_t)blockIdx.x * kTraceBlock + threadIdx.x;
  WS stk;
  stk.bind(smem + (size_t)wave * WS::kWaveBytes, lane);
  stk.overflow = sc.wstack_overflow ? sc.wstack_overflow + slot * sc.woverflow_cap : nullptr;
  if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0ull;
  __syncthreads();

  // wave-uniform cursor over ray indices
  uint32_t cur_next = 0, cur_end = 0;
  bool exhausted = false;
  // per-lane state
  int st = TS_EMIT;
  bool have_ray = false;
  uint32_t rid = 0;
  V3 org = v3(0, 0, 0), dir = v3(0, 0, 1);
  double ix = 0, iy = 0, iz = 0;
  uint32_t sgn = 0; // bit k: dir[k] < 0
  bool ray_plain = false; // the ray may take the min/max form of the slab test (mgpu_device.hpp, slab_hit)
  int sp = 0;               // far children on the stack
  uint32_t cur = kWNone;    // wide record to enter next (kWNone: pop)
  double bt = kDblMax, bu = 0, bv = 0;
  uint32_t bslot = kNoHit;
  uint32_t tri_cur = 0, tri_end = 0;
  uint32_t n_rays = 0, n_nodes = 0, n_tris = 0;

  for (;;) {
    const unsigned long long mN = MGPU_BALLOT(st == TS_NODE);
    const unsigned long long mT = MGPU_BALLOT(st == TS_TRI);
    const unsigned long long mE = MGPU_BALLOT(st == TS_EMIT);
    const int cN = __popcll(mN), cT = __popcll(mT), cE = __popcll(mE);
    if ((cN | cT | cE) == 0) break;
    const bool run_emit = (cE >= MGPU_EMIT_MIN) || (cN == 0 && cT == 0);
    if (!run_emit && cN * MGPU_TRACE_NODE_WEIGHT >= cT) {
      // ================================ NODE step ================================
      const bool all_plain = MGPU_BALLOT(st == TS_NODE && !ray_plain) == 0ull; // wave-uniform
      if (st == TS_NODE) {
        const bool sx = (sgn & 1u) != 0u, sy = (sgn & 2u) != 0u, sz = (sgn & 4u) != 0u;
        // slab test in min/max form when every lane's ray qualifies, the literal form for this step otherwise
        int r;
        if (all_plain)
          r = wide_node_step<true, 4, kWideStackLds>(sc.wnodes, stk, org, ix, iy, iz, sx, sy, sz, sgn, bt, cur, sp, tri_cur, tri_end, n_nodes);
        else
          r = wide_node_step<false, 4, kWideStackLds>(sc.wnodes, stk, org, ix, iy, iz, sx, sy, sz, sgn, bt, cur, sp, tri_cur, tri_end, n_nodes);
        if (r == WT_TRI) st = TS_TRI;
        else if (r == WT_DONE) st = TS_EMIT;
      }
    } else if (!run_emit) {
      // ================================ TRI step =================================
      bool shared_done = false;
      if (cT <= 32) { // 2 or 4 lanes per open leaf (mgpu_device.hpp, shared_leaves_step)
        uint32_t my_trips = 0;
        shared_done = shared_leaves_step<false, 16>(mT, cT, cT <= 16 ? 2 : 1, lane, s_owner + wave * 64, st == TS_TRI, nullptr, sc.tris,
                                                    org, dir, tri_cur, tri_end, bt, bu, bv, bslot, n_tris, my_trips);
      }
      if (!shared_done && st == TS_TRI) {
#pragma unroll 1
        for (int rep = 0; rep < 16; ++rep) {
          const DTri *tp = sc.tris + tri_cur;
          const double2 a0 = reinterpret_cast<const double2 *>(tp)[0];
          const double2 a1 = reinterpret_cast<const double2 *>(tp)[1];
          const double2 a2 = reinterpret_cast<const double2 *>(tp)[2];
          const double2 a3 = reinterpret_cast<const double2 *>(tp)[3];
          const double e2z = tp->e2[2];
          ++n_tris;
          // TriangleIsect, bvh_accel.cc:595-638
          const V3 p0 = v3(a0.x, a0.y, a1.x), e1 = v3(a1.y, a2.x, a2.y), e2 = v3(a3.x, a3.y, e2z);
          const V3 p = cross(dir, e2);
          const double det = dot(e1, p);
          if (!(fabs(det) < kDblEps1024)) {
            const double invDet = inv_det_w(det); // 1.0 / det
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
      }
      if (st == TS_TRI && tri_cur == tri_end) st = (sp == 0) ? TS_EMIT : TS_NODE; // cur == kWNone: the next NODE step pops
    } else {
      // ================================ EMIT step ================================
      const bool emit_lane = (st == TS_EMIT);
      const bool emitting = emit_lane && have_ray;
      // a miss leaves t = DBL_MAX, u = v = 0, faceID = -1 (bvh_accel.cc:782-786); every other field is zeroed here.
      // Traverse reports a hit iff isect.t < DBL_MAX (bvh_accel.cc:838): a NaN t (NaN ray) fails that test even though
      // TestLeafNode accepted a triangle and already wrote faceID / materialID.
      const bool hit = bt < kDblMax;
      uint32_t faceID = 0xFFFFFFFFu, materialID = 0, f0 = 0, f1 = 0, f2 = 0;
      V3 pos = v3(0, 0, 0), gn = v3(0, 0, 0), sn = v3(0, 0, 0);
      double tc0 = 0.0, tc1 = 0.0;
      if (emitting) {
        if (!hit && bslot != kNoHit) {
          faceID = sc.tris[bslot].face;
          materialID = sc.tris[bslot].mat;
        }
        if (hit) {
          // BuildIntersection, bvh_accel.cc:699-769
          const DTri *tp = sc.tris + bslot;
          const uint32_t face = tp->face;
          faceID = face;
          materialID = tp->mat;
          f0 = sc.faces[3 * (size_t)face + 0];
          f1 = sc.faces[3 * (size_t)face + 1];
          f2 = sc.faces[3 * (size_t)face + 2];
          pos = v3(org.x + bt * dir.x, org.y + bt * dir.y, org.z + bt * dir.z);
          const V3 e1 = v3(tp->e1[0], tp->e1[1], tp->e1[2]), e2 = v3(tp->e2[0], tp->e2[1], tp->e2[2]);
          gn = normalized(cross(e1, e2));
          if (sc.fv_normals) {
            const double *nn = sc.fv_normals + 9 * (size_t)face;
            const double w = 1.0 - bu - bv;
            sn = v3(w * nn[0] + bu * nn[3] + bv * nn[6], w * nn[1] + bu * nn[4] + bv * nn[7],
                    w * nn[2] + bu * nn[5] + bv * nn[8]);
          } else {
            sn = gn;
          }
          if (sc.fv_uvs) {
            const double *uv = sc.fv_uvs + 6 * (size_t)face;
            const double w = 1.0 - bu - bv;
            tc0 = w * uv[0] + bu * uv[2] + bv * uv[4];
            tc1 = w * uv[1] + bu * uv[3] + bv * uv[5];
          }
        }
        hit_out[rid] = hit ? 1 : 0;
      }
      // The 184-byte records go out through LDS, 16 at a time: their lanes lay them out in the wave's staging area and
      // the whole wave stores them as 8-byte pieces, neighbouring lanes writing neighbouring pieces of one record -- a
      // lane storing its own record would issue 23 store instructions of one isolated 8-byte write per lane each.
      {
        const unsigned long long em = MGPU_BALLOT(emitting);
        const int total = __popcll(em);
        const int my_rank = (int)__popcll(em & ((1ull << lane) - 1ull));
        unsigned long long *stage = s_stage + (size_t)wave * (16 * 23 + 16);
        uint32_t *stage_rid = reinterpret_cast<uint32_t *>(stage + 16 * 23);
        for (int r0 = 0; r0 < total; r0 += 16) {
          if (emitting && my_rank >= r0 && my_rank < r0 + 16) {
            MgpuIntersection *is = reinterpret_cast<MgpuIntersection *>(stage + (size_t)(my_rank - r0) * 23);
            is->t = bt; is->u = bu; is->v = bv;
            is->faceID = faceID; is->materialID = materialID; is->f0 = f0; is->f1 = f1; is->f2 = f2; is->pad_ = 0;
            is->position[0] = pos.x; is->position[1] = pos.y; is->position[2] = pos.z;
            is->geometricNormal[0] = gn.x; is->geometricNormal[1] = gn.y; is->geometricNormal[2] = gn.z;
            is->normal[0] = sn.x; is->normal[1] = sn.y; is->normal[2] = sn.z;
            for (int k = 0; k < 3; ++k) { is->tangent[k] = 0.0; is->binormal[k] = 0.0; }
            is->texcoord[0] = tc0; is->texcoord[1] = tc1;
            stage_rid[my_rank - r0] = rid;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int cnt = (total - r0 < 16) ? (total - r0) : 16;
          for (int i = lane; i < cnt * 23; i += 64) {
            const int rec = i / 23, piece = i - rec * 23;
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(out + stage_rid[rec]) + piece;
            *dst = stage[i];
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier(); // the area is rewritten by the next 16
        }
      }
      if (emitting) have_ray = false;
      // ---- hand-out of ray indices, executed by the whole wave (the cursor is wave-uniform) ----
      bool want = emit_lane;
      for (;;) {
        const unsigned long long wm = MGPU_BALLOT(want);
        if (!wm || exhausted) break;
        if (cur_next >= cur_end) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(work_counter, kRayChunk);
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          if (base >= n) { exhausted = true; break; }
          cur_next = base;
          cur_end = (n - base < kRayChunk) ? n : base + kRayChunk;
        }
        const uint32_t rank = __popcll(wm & ((1ull << lane) - 1ull));
        const uint32_t avail = cur_end - cur_next;
        if (want && rank < avail) {
          rid = cur_next + rank;
          have_ray = true;
          want = false;
        }
        const uint32_t took = min((uint32_t)__popcll(wm), avail);
        cur_next += took;
      }
      if (emit_lane) {
        if (have_ray) {
          // BVHAccel::Traverse prologue, bvh_accel.cc:774-802 (only org / dir of the 88-byte Ray are inputs)
          const MgpuRay *r = rays + rid;
          org = v3(r->org[0], r->org[1], r->org[2]);
          dir = v3(r->dir[0], r->dir[1], r->dir[2]);
          sgn = (dir.x < 0.0 ? 1u : 0u) | (dir.y < 0.0 ? 2u : 0u) | (dir.z < 0.0 ? 4u : 0u);
          const bool inv_ok = inverse_dir_w(dir, ix, iy, iz); // 1.0 / dir, no zero guard, as the reference
          ray_plain = sc.boxes_ordered && inv_ok && origin_is_finite(org);
          bt = kDblMax; bu = 0.0; bv = 0.0; bslot = kNoHit;
          sp = 0;
          cur = sc.wroot; // the super root: its child 0 is the tree's root (whose box test is the reference's first pop)
          n_nodes -= 1u;  // ... and its child 1 a dummy the reference never pops
          ++n_rays;
          st = TS_NODE;
        } else {
          st = TS_IDLE;
        }
      }
    }
  }

  // counters: wave reduction, one LDS atomic per wave, one global atomic per workgroup and word
  unsigned long long rn = n_rays, nn = n_nodes, tn = n_tris;
  for (int off = 32; off; off >>= 1) {
    rn += __shfl_down(rn, off);
    nn += __shfl_down(nn, off);
    tn += __shfl_down(tn, off);
  }
  if (lane == 0) {
    atomicAdd(&s_cnt[0], rn);
    atomicAdd(&s_cnt[1], nn);
    atomicAdd(&s_cnt[2], tn);
  }
  __syncthreads();
  if (threadIdx.x == 0 && stats) {
    atomicAdd(&stats[kStatRays], s_cnt[0]);
    atomicAdd(&stats[kStatNodes], s_cnt[1]);
    atomicAdd(&stats[kStatTris], s_cnt[2]);
    atomicAdd(&stats[kStatTraceCalls], s_cnt[0]);
  }
}

hipError_t launch_trace_sm(dim3 grid, hipStream_t s, const DScene &sc, const MgpuRay *rays, uint32_t n, MgpuIntersection *out,
                           uint8_t *hit, uint32_t *counter, unsigned long long *stats, const uint32_t *select) {
  const size_t shmem = (size_t)(kTraceBlock / 64) * (WStack<kWideStackLds>::kWaveBytes + (16 * 23 + 16) * sizeof(unsigned long long));
  hipLaunchKernelGGL(k_trace_sm, grid, dim3(kTraceBlock), shmem, s, sc, rays, n, out, hit, counter, stats, select);
  return hipGetLastError();
}

} // namespace mgpu
