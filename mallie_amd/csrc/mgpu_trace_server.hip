// mgpu_trace_server.hip -- a resident traversal kernel for callers that bring ONE ray per call (Scene::Trace, scene.cc:253-315,
// called per ray from every OpenMP thread of the reference: render.cc:403).
//
// A launch per call costs ~22 us of launch + completion round trip however short the kernel is (profiles/experiments/README.md,
// round 3).  This kernel takes the launch out of the call: while callers are around, a few waves stay resident and poll a
// MAILBOX in host memory the device maps (TraceMailbox, mgpu_kernels.hpp).  A caller owns a slot for the duration of its call,
// writes the ray, publishes a sequence number; the wave that owns the slot sees the number, walks the ray with the very
// traverse() the batched k_trace uses, writes the 184-byte Intersection straight into the slot's host memory and publishes the
// number back.  No launch, no copy engine, no stream synchronisation in a call: two PCIe crossings and one traversal.
//
// The kernel never outlives its use: every wave leaves when the server has been idle for `idle_ticks`, when the host raises
// `stop` (the render entry points do, they want every CU), or when the launch is `life_ticks` old -- all on the constant
// 100 MHz wall clock, plus a hard cap on the number of polls, so it cannot spin for ever whatever the host does.  Leaving is
// collective: the first wave to see a reason raises `quit` in device memory, every wave checks it before it polls, and the
// last one out publishes the launch's epoch in `exited_epoch`.  A request that arrives too late for a launch is simply still
// pending in the mailbox: the caller sees the epoch, launches the next server, and that one starts from the `ack` numbers it
// finds.  Every word of the mailbox has one writer (host: req, ray, stop; device: ack, hit, rec, exited_epoch).
#include "mgpu_device.hpp"
#include "mgpu_kernels.hpp"

namespace mgpu {

__device__ __forceinline__ uint32_t sys_load(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store(uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// BVHAccel::Traverse for the requests a wave has found, at most one per lane (bvh_accel.cc:773-844, without BuildIntersection).
// The node walk is traverse()'s, lane by lane (mgpu_device.hpp: same pops, same pushes, same box test).  The leaves are where a
// lone ray loses its time -- TestLeafNode walks a leaf's triangles one after the other, each a dependent fetch + ~60 dependent
// fp64 operations, and most of a server wave's lanes have nothing to do -- so an open leaf is served by the WHOLE wave: its
// owner's ray is broadcast, lane i evaluates TriangleIsect's arithmetic for the leaf's i-th triangle up to, not including, the
// comparison with the best t, and the accept rule is then replayed over the candidates IN LEAF ORDER with wave-uniform values:
// `!(t > best)` accepts, exactly as the reference's `if (t < 0.0 || t > tBest) continue` leaves it -- ties go to the later
// triangle, a NaN t is accepted and makes every later candidate accepted.  Same hits, same counters, one step per leaf.
template <int CAP>
__device__ __forceinline__ void traverse_coop(const DScene &sc, const Stack<CAP, true> &stk, int lane, bool active, V3 org, V3 dir,
                                              Hit &h, Counters &c) {
  const bool sx = dir.x < 0.0, sy = dir.y < 0.0, sz = dir.z < 0.0;
  const uint32_t sgn = (sx ? 1u : 0u) | (sy ? 2u : 0u) | (sz ? 4u : 0u);
  double ix, iy, iz;
  const bool inv_ok = inverse_dir_w(dir, ix, iy, iz);
  const bool all_plain = MGPU_BALLOT(!(sc.boxes_ordered && inv_ok && origin_is_finite(org))) == 0ull;
  h.t = kDblMax; h.u = 0.0; h.v = 0.0; h.slot = kNoHit;
  int sp = -1;
  if (active) {
    sp = 0;
    stk.put(0, 0u);
  }
  uint32_t leaf_first = 0, leaf_cnt = 0;
  uint32_t nnodes = 0, ntris = 0;
  for (;;) {
    while (sp >= 0 && leaf_cnt == 0) { // traverse()'s node loop
      const uint32_t ni = stk.get(sp);
      --sp;
      ++nnodes;
      const MgpuNode *nd = sc.nodes + ni;
      const double2 b0 = *reinterpret_cast<const double2 *>(&nd->bmin[0]);
      const double2 b1 = *reinterpret_cast<const double2 *>(&nd->bmin[2]);
      const double2 b2 = *reinterpret_cast<const double2 *>(&nd->bmax[1]);
      int4 meta = *reinterpret_cast<const int4 *>(&nd->flag);
      MGPU_KEEP4(meta.x, meta.y, meta.z, meta.w);
      const bool hit = all_plain ? slab_hit<true>(b0, b1, b2, org, ix, iy, iz, sx, sy, sz, h.t)
                                 : slab_hit<false>(b0, b1, b2, org, ix, iy, iz, sx, sy, sz, h.t);
      if (hit) {
        if (meta.x == 0) {
          const bool nearIsSecond = ((sgn >> (uint32_t)meta.y) & 1u) != 0u;
          const uint32_t c0 = (uint32_t)meta.z, c1 = (uint32_t)meta.w;
          stk.put(sp + 1, nearIsSecond ? c0 : c1);
          stk.put(sp + 2, nearIsSecond ? c1 : c0);
          sp += 2;
        } else {
          leaf_cnt = (uint32_t)meta.z;
          leaf_first = (uint32_t)meta.w;
        }
      }
    }
    unsigned long long open = MGPU_BALLOT(leaf_cnt != 0);
    if (!open) break; // every lane has run dry
    while (open) {    // one open leaf at a time, its triangles across the lanes
      const int L = __ffsll((long long)open) - 1;
      open &= open - 1;
      const V3 o = v3(__shfl(org.x, L), __shfl(org.y, L), __shfl(org.z, L));
      const V3 d = v3(__shfl(dir.x, L), __shfl(dir.y, L), __shfl(dir.z, L));
      const uint32_t first = __shfl(leaf_first, L), cnt = __shfl(leaf_cnt, L);
      double bt = __shfl(h.t, L), bu = __shfl(h.u, L), bv = __shfl(h.v, L);
      uint32_t bslot = __shfl(h.slot, L);
      for (uint32_t base = 0; base < cnt; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        bool cand = false;
        double t = 0.0, u = 0.0, v = 0.0;
        if (i < cnt) { // TriangleIsect, bvh_accel.cc:595-638, up to the comparison with the best t
          const DTri *tp = sc.tris + (first + i);
          const double2 a0 = reinterpret_cast<const double2 *>(tp)[0];
          const double2 a1 = reinterpret_cast<const double2 *>(tp)[1];
          const double2 a2 = reinterpret_cast<const double2 *>(tp)[2];
          const double2 a3 = reinterpret_cast<const double2 *>(tp)[3];
          const double e2z = tp->e2[2];
          const V3 p0 = v3(a0.x, a0.y, a1.x), e1 = v3(a1.y, a2.x, a2.y), e2 = v3(a3.x, a3.y, e2z);
          const V3 p = cross(d, e2);
          const double det = dot(e1, p);
          if (!(fabs(det) < kDblEps1024)) {
            const double invDet = inv_det_w(det);
            const V3 s = o - p0;
            const V3 q = cross(s, e1);
            u = dot(s, p) * invDet;
            v = dot(q, d) * invDet;
            t = dot(e2, q) * invDet;
            cand = !(u < 0.0 || u > 1.0) && !(v < 0.0 || u + v > 1.0) && !(t < 0.0);
          }
        }
        unsigned long long pm = MGPU_BALLOT(cand);
        while (pm) { // the reference's loop over the leaf, for the triangles that got as far as `t > tBest`
          const int j = __ffsll((long long)pm) - 1;
          pm &= pm - 1;
          const double tj = __shfl(t, j);
          if (!(tj > bt)) {
            bt = tj;
            bu = __shfl(u, j);
            bv = __shfl(v, j);
            bslot = first + base + (uint32_t)j;
          }
        }
      }
      if (lane == L) {
        h.t = bt; h.u = bu; h.v = bv; h.slot = bslot;
        ntris += cnt;
        leaf_cnt = 0;
      }
    }
  }
  c.nodes += nnodes;
  c.tris += ntris;
  c.rays += active ? 1u : 0u;
}

template <int CAP>
__global__ __launch_bounds__(64) void k_trace_server(DScene sc, TraceMailbox *mb, TraceServerCtl *ctl, uint32_t epoch,
                                                     unsigned long long idle_ticks, unsigned long long life_ticks,
                                                     unsigned long long max_polls, uint32_t stage_nodes_bytes,
                                                     uint32_t stage_tris_bytes) {
  __shared__ __attribute__((aligned(16))) uint32_t s_stack[CAP][64];
  MGPU_DYN_SHARED(unsigned char, s_scene); // nodes, then triangles, when the scene is small enough
  const int lane = threadIdx.x;
  const int wave = blockIdx.x;
  const bool owner = lane < kSrvSlotsPerWave;
  const int slot = wave * kSrvSlotsPerWave + (owner ? lane : 0);
  Stack<CAP, true> stk;
  stk.lds = &s_stack[0][lane];
  stk.overflow = sc.stack_overflow ? sc.stack_overflow + ((size_t)wave * 64 + lane) * sc.overflow_cap : nullptr;
  // A scene that fits is copied into this workgroup's LDS once per launch (cornellbox_suzanne: 13 + 78 KB): a lone ray's walk is a
  // chain of dependent fetches, and an LDS fetch is a fraction of an L2 one.  The walk reads it through generic pointers.
  if (stage_nodes_bytes) {
    const uint4 *src_n = reinterpret_cast<const uint4 *>(sc.nodes), *src_t = reinterpret_cast<const uint4 *>(sc.tris);
    uint4 *dst_n = reinterpret_cast<uint4 *>(s_scene), *dst_t = reinterpret_cast<uint4 *>(s_scene + stage_nodes_bytes);
    for (uint32_t i = lane; i < stage_nodes_bytes / 16; i += 64) dst_n[i] = src_n[i];
    for (uint32_t i = lane; i < stage_tris_bytes / 16; i += 64) dst_t[i] = src_t[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    sc.nodes = reinterpret_cast<const MgpuNode *>(s_scene);
    sc.tris = reinterpret_cast<const DTri *>(s_scene + stage_nodes_bytes);
  }
  // what this slot has been served up to: the previous launch's last acknowledgement (one writer: the device)
  uint32_t served = owner ? sys_load(&mb->ack[slot]) : 0u;
  const unsigned long long t_start = wall_clock64();
  for (unsigned long long poll = 0; poll < max_polls; ++poll) {
    if (MGPU_WAVE_LOAD(__hip_atomic_load(&ctl->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break; // (the wave leaves together)
    // one poll: lanes 0..15 read their slots' request numbers (one 64-byte line of host memory), lane 16 the stop word
    // (ONE load instruction for both: two were two PCIe round trips per poll)
    uint32_t r = served;
    const uint32_t *word = owner ? &mb->req[slot] : &mb->stop;
    if (lane <= kSrvSlotsPerWave) r = sys_load(word);
    const bool work = owner && r != served;
    const bool stop = __shfl(r, kSrvSlotsPerWave) != 0u;
    const unsigned long long now = wall_clock64();
    if (MGPU_BALLOT(work)) {
      V3 org = v3(0.0, 0.0, 0.0), dir = v3(1.0, 1.0, 1.0); // lanes without a request carry a harmless ray and never start it
      if (work) {
        // the ray: three 16-byte loads in flight together (ONE PCIe round trip; the request number's acquire orders them behind
        // it, and mapped host memory is not cached on the device).  Six system-scope atomic loads were six round trips: 7 us.
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const d2_t *rp = reinterpret_cast<const d2_t *>(&mb->ray[slot][0]);
        const d2_t r0 = __builtin_nontemporal_load(rp), r1 = __builtin_nontemporal_load(rp + 1), r2 = __builtin_nontemporal_load(rp + 2);
        org = v3(r0.x, r0.y, r1.x);
        dir = v3(r1.y, r2.x, r2.y);
      }
      Hit h;
      Counters c{};
      traverse_coop<CAP>(sc, stk, lane, work, org, dir, h, c);
      if (work) {
        const bool hit = h.t < kDblMax; // bvh_accel.cc:838
#ifdef MGPU_SRV_PROFILE
        mb->prof[slot][1] = (uint32_t)(wall_clock64() - now);
        mb->prof[slot][2] = c.nodes;
        mb->prof[slot][3] = c.tris;
#endif
        fill_intersection(sc, org, dir, h, hit, &mb->rec[slot]);
        mb->hit[slot] = hit ? 1u : 0u;
        mb->ticks[slot] = (uint32_t)(wall_clock64() - now);
        sys_store(&mb->ack[slot], r); // release at system scope: the record is visible to the host before the number
        served = r;
      }
      if (lane == 0) __hip_atomic_fetch_max(&ctl->last_work, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (lane == 0) {
      unsigned long long last = __hip_atomic_load(&ctl->last_work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (last < t_start) last = t_start;
      // (`last` may be another wave's later clock reading -- fetch_max above -- in which case the unsigned difference would wrap)
      if (stop || (last < now && now - last > idle_ticks)) __hip_atomic_store(&ctl->quit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0 && now - t_start > life_ticks) __hip_atomic_store(&ctl->quit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (lane == 0) {
    __hip_atomic_store(&ctl->quit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // a wave at its poll cap takes the others with it
    const uint32_t before = __hip_atomic_fetch_add(&ctl->exited, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1 == gridDim.x) sys_store(&mb->exited_epoch, epoch); // every acknowledgement of this launch is out
  }
}

template <int CAP>
static hipError_t launch_server_cap(hipStream_t s, const DScene &sc, TraceMailbox *mb, TraceServerCtl *ctl, uint32_t epoch,
                                    unsigned long long idle_ticks, unsigned long long life_ticks, unsigned long long max_polls,
                                    uint32_t stage_nodes_bytes, uint32_t stage_tris_bytes) {
  const size_t shmem = (size_t)stage_nodes_bytes + stage_tris_bytes;
  if (shmem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_trace_server<CAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_trace_server<CAP>, dim3(kSrvWaves), dim3(64), shmem, s, sc, mb, ctl, epoch, idle_ticks, life_ticks, max_polls,
                     stage_nodes_bytes, stage_tris_bytes);
  return hipGetLastError();
}

hipError_t launch_trace_server(int cap, hipStream_t s, const DScene &sc, TraceMailbox *mb, TraceServerCtl *ctl, uint32_t epoch,
                               unsigned long long idle_ticks, unsigned long long life_ticks, unsigned long long max_polls,
                               uint32_t stage_nodes_bytes, uint32_t stage_tris_bytes) {
  switch (cap) {
  case 16: return launch_server_cap<16>(s, sc, mb, ctl, epoch, idle_ticks, life_ticks, max_polls, stage_nodes_bytes, stage_tris_bytes);
  case 24: return launch_server_cap<24>(s, sc, mb, ctl, epoch, idle_ticks, life_ticks, max_polls, stage_nodes_bytes, stage_tris_bytes);
  default: return launch_server_cap<32>(s, sc, mb, ctl, epoch, idle_ticks, life_ticks, max_polls, stage_nodes_bytes, stage_tris_bytes);
  }
}

} // namespace mgpu
