// mgpu_kernels.hip -- gfx950 kernels of the render hot path (wave64, fp64, no MFMA: branchy scalar traversal).
//
//   k_trace  : batched Scene::Trace  (scene.cc:253 -> BVHAccel::Traverse bvh_accel.cc:773 + BuildIntersection :699)
//   k_render : persistent-threads path tracer = Render()'s pixel loop (render.cc:657-681) + PathTrace (render.cc:381-456)
//
// Compiled with -ffp-contract=off (see mgpu_device.hpp for the arithmetic contract).
#include "mgpu_device.hpp"
#include "mgpu_kernels.hpp"

namespace mgpu {

// =====================================================================================================================
// k_trace: one lane per ray, a wave walks the ray array in strides of the grid (64 consecutive rays per trip).
// The 184-byte Intersection records of a wave's 64 rays are contiguous in memory: the lanes assemble them in LDS and
// the wave writes the 11 776 bytes out with full-width stores (a lane storing its own record would touch 64 different
// cache lines per store instruction).  Counters are reduced per workgroup before the one atomic per word.
// =====================================================================================================================
constexpr int kIsectWords = (int)(sizeof(MgpuIntersection) / 4); // 46

template <int CAP>
__global__ __launch_bounds__(kBlock) void k_trace(DScene sc, const MgpuRay *__restrict__ rays, size_t n,
                                                 MgpuIntersection *__restrict__ out, uint8_t *__restrict__ hit_out,
                                                 unsigned long long *__restrict__ stats,
                                                 const uint32_t *__restrict__ select) {
  if (select && *select != kTraceSelectV1) return; // k_trace_probe chose the other kernel for this batch
  __shared__ __attribute__((aligned(16))) uint32_t s_stack[kBlock / 64][CAP][64];
  __shared__ unsigned long long s_cnt[3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t slot = (size_t)blockIdx.x * kBlock + threadIdx.x; // hardware lane slot (stack overflow column)
  Stack<CAP, true> stk;
  stk.lds = &s_stack[wave][0][lane];
  stk.overflow = sc.stack_overflow ? sc.stack_overflow + slot * sc.overflow_cap : nullptr;
  if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0ull;
  __syncthreads();
  Counters c{};
  const size_t wave_stride = (size_t)gridDim.x * (kBlock / 64) * 64;
  for (size_t base = ((size_t)blockIdx.x * (kBlock / 64) + wave) * 64; base < n; base += wave_stride) {
    const size_t gid = base + lane;
    const bool live = gid < n;
    V3 org = v3(0, 0, 0), dir = v3(0, 0, 1);
    Hit h;
    h.t = kDblMax; h.u = 0.0; h.v = 0.0; h.slot = kNoHit;
    if (live) {
      const MgpuRay *r = rays + gid;
      org = v3(r->org[0], r->org[1], r->org[2]);
      dir = v3(r->dir[0], r->dir[1], r->dir[2]);
      traverse<CAP, true>(sc, stk, org, dir, h, c);
    }
    // Traverse reports a hit iff isect.t < DBL_MAX (bvh_accel.cc:838): a NaN t (NaN ray) fails that test even though
    // TestLeafNode accepted a triangle and already wrote faceID / materialID.
    const bool hit = h.t < kDblMax;
    // The wave's traversal stacks are idle now: 16 records at a time are assembled in that LDS area (16 * 184 B = 2944 B
    // <= CAP * 256 B) by their lanes and streamed out by all 64 lanes.
    uint32_t *stage = &s_stack[wave][0][0];
    const size_t cnt = (n - base < 64) ? (n - base) : 64;
    // every lane's traversal has left the stacks before the first record is written over them (the lanes reconverge here anyway;
    // said explicitly -- the wave barrier costs no instruction -- and needed by the tests' wave emulator, whose lanes do not run in lockstep)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int q = 0; q < 4; ++q) {
      if (live && (lane >> 4) == q) {
        MgpuIntersection *is = reinterpret_cast<MgpuIntersection *>(stage + (lane & 15) * kIsectWords);
        fill_intersection(sc, org, dir, h, hit, is);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int have = (int)cnt - 16 * q; // records of this quarter that exist
      uint32_t *q_out = reinterpret_cast<uint32_t *>(out + base + 16 * q); // 16-byte aligned: 16 records = 2944 B
      if (have >= 16) {
        const uint4 *l4 = reinterpret_cast<const uint4 *>(stage);
        uint4 *o4 = reinterpret_cast<uint4 *>(q_out);
        for (int i = lane; i < 16 * kIsectWords / 4; i += 64) o4[i] = l4[i];
      } else if (have > 0) {
        for (int i = lane; i < have * kIsectWords; i += 64) q_out[i] = stage[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier(); // the area is rewritten by the next quarter / the next trip's stacks
    }
    if (live) hit_out[gid] = hit ? 1 : 0;
  }
  // counters: wave reduction, then one LDS atomic per wave and one global atomic per workgroup and word
  unsigned long long rn = c.rays, nn = c.nodes, tn = c.tris;
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

// =====================================================================================================================
// k_render: persistent waves; every lane owns one pixel at a time for all its passes, then takes the next pixel.
// =====================================================================================================================
//
// Work distribution: the pixel set is cut into 8x8 tiles (one wave-fetch = one tile = 64 pixels, so the lanes of a
// freshly fed wave shoot coherent primary rays); tiles are handed out in chunks of kChunkTiles by one global counter
// per launch (a few thousand atomics per frame).  A lane that finishes its pixel (all passes) early pulls the next
// pixel of the wave's chunk right away, so path-length divergence does not idle lanes ("path regeneration").
//
// Float accumulation order: a pixel's passes are summed in pass order in float32 by the lane that owns it, which is
// exactly Render() + AccumImage (main_sdl.cc:138-143); no atomics touch the image.
enum : int { S_NEED_PIXEL = 0, S_NEED_PATH = 1, S_TRACE = 2 };

#ifndef MGPU_RENDER_MIN_WAVES
#define MGPU_RENDER_MIN_WAVES 1
#endif
template <int CAP>
__global__ __launch_bounds__(kBlock, MGPU_RENDER_MIN_WAVES) void k_render(DScene sc, RenderParams P) {
  __shared__ uint32_t s_stack[kBlock / 64][CAP][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  Stack<CAP, true> stk;
  stk.lds = &s_stack[wave][0][lane];
  stk.overflow = sc.stack_overflow ? sc.stack_overflow + gid * sc.overflow_cap : nullptr;

  const int win_w = P.x1 - P.x0;
  const uint32_t tiles_x = (uint32_t)(win_w + 7) >> 3;
  const uint32_t tiles_y = (uint32_t)(P.n_rows + 7) >> 3;
  const uint32_t total_tiles = tiles_x * tiles_y;

  // wave-uniform chunk of tiles [tile_next, tile_end); pixel cursor inside the current tile
  uint32_t tile_next = 0, tile_end = 0, in_tile = 64;
  bool exhausted = false;

  int state = S_NEED_PIXEL;
  uint32_t lx = 0, ly = 0; // local pixel (window column, local row)
  int pass = 0;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  Rng rng{1, 0, 0, 0};
  V3 org = v3(0, 0, 0), dir = v3(0, 0, 1);
  double thr0 = 1, thr1 = 1, thr2 = 1, rad0 = 0, rad1 = 0, rad2 = 0;
  int pathLength = 1;
  uint32_t last_mat = kNoMaterial; // Intersection::materialID as the reference would still hold it (stale on a miss)
  Counters c{};
  uint32_t trace_calls = 0, paths = 0;
#ifdef MGPU_UTIL
  uint32_t u_outer = 0, u_trace = 0, u_shade = 0, u_gen = 0;
#endif
  bool probe_on = false;

  for (;;) {
    // ---- 1. hand pixels to idle lanes ---------------------------------------------------------------------------
    {
      const unsigned long long need = MGPU_BALLOT(state == S_NEED_PIXEL);
      if (need) {
        if (in_tile >= 64 && !exhausted) { // current tile used up: advance (refill the chunk when empty)
          if (tile_next >= tile_end) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(P.work_counter, (uint32_t)kChunkTiles);
            base = __shfl(base, 0);
            tile_next = base;
            tile_end = min(base + (uint32_t)kChunkTiles, total_tiles);
            if (base >= total_tiles) { exhausted = true; tile_end = tile_next = total_tiles; }
          }
          if (!exhausted) { in_tile = 0; }
        }
        if (!exhausted && in_tile < 64) {
          const uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
          const uint32_t slot = in_tile + rank;
          if (state == S_NEED_PIXEL && slot < 64) {
            const uint32_t tx = tile_next % tiles_x, ty = tile_next / tiles_x;
            const uint32_t x = tx * 8 + (slot & 7), y = ty * 8 + (slot >> 3);
            if (x < (uint32_t)win_w && y < (uint32_t)P.n_rows) {
              lx = x; ly = y; pass = 0;
              acc0 = acc1 = acc2 = 0.f;
              state = S_NEED_PATH;
            } // pixels of a partial edge tile that fall outside the window are simply skipped
          }
          in_tile += (uint32_t)__popcll(need);
          if (in_tile >= 64) { in_tile = 64; ++tile_next; }
        }
      }
      if (__all(state == S_NEED_PIXEL)) {
        if (exhausted) break;
        continue;
      }
    }

#ifdef MGPU_UTIL
    u_outer += 1;
    u_gen += (uint32_t)__popcll(MGPU_BALLOT(state == S_NEED_PATH));
#endif
    // ---- 2. start a new eye path (PathTrace prologue, render.cc:387-400) ----------------------------------------
    if (state == S_NEED_PATH) {
      const uint32_t j = ly;
      const int gy = P.y_first + (int)(j / (uint32_t)P.strip_h) * P.y_period + (int)(j % (uint32_t)P.strip_h);
      const int gx = P.x0 + (int)lx;
      const uint32_t gpix = (uint32_t)gy * (uint32_t)P.W + (uint32_t)gx;
      uint32_t st[4];
      if (P.rng_mode == MGPU_RNG_TABLE) {
        const uint4 s4 = reinterpret_cast<const uint4 *>(P.rng_states)[(size_t)pass * P.W * P.H + gpix];
        st[0] = s4.x; st[1] = s4.y; st[2] = s4.z; st[3] = s4.w;
      } else {
        hash_state(P.seed, P.pass_base + (uint32_t)pass, gpix, st);
      }
      rng = Rng{st[0], st[1], st[2], st[3]};
      probe_on = P.probe && gpix == P.probe_pixel && (uint32_t)pass == P.probe_pass;
      const float ju = (float)(rng_next(rng) - 0.5);
      const float jv = (float)(rng_next(rng) - 0.5);
      org = v3(P.frame[0], P.frame[1], P.frame[2]);
      dir = camera_dir(P.frame, (double)((float)gx + ju), (double)((float)gy + jv));
      thr0 = thr1 = thr2 = 1.0;
      rad0 = rad1 = rad2 = 0.0;
      pathLength = 1;
      ++paths;
      state = S_TRACE;
    }

    // ---- 3. Scene::Trace for every lane that holds a ray --------------------------------------------------------
    Hit h;
    h.slot = kNoHit;
    h.t = kDblMax;
    h.u = h.v = 0.0;
#ifdef MGPU_UTIL
    u_trace += (uint32_t)__popcll(MGPU_BALLOT(state == S_TRACE));
#endif
    if (state == S_TRACE) traverse<CAP, true>(sc, stk, org, dir, h, c);

    // ---- 4. the rest of one PathTrace loop iteration (render.cc:403-452) ----------------------------------------
    if (state == S_TRACE) {
      bool hit = h.t < kDblMax; // bvh_accel.cc:838
      double t = h.t;
      V3 n = v3(0, 0, 0);
      if (h.slot != kNoHit) last_mat = sc.tris[h.slot].mat; // written by TestLeafNode on every accepted triangle
      if (hit) {
        if (sc.has_fv_normals) { // barycentric lerp, not renormalised (bvh_accel.cc:745-748)
          const double *nn = sc.slot_normal + 9 * (size_t)h.slot;
          const double w = 1.0 - h.u - h.v;
          n.x = w * nn[0] + h.u * nn[3] + h.v * nn[6];
          n.y = w * nn[1] + h.u * nn[4] + h.v * nn[7];
          n.z = w * nn[2] + h.u * nn[5] + h.v * nn[8];
        } else {
          const double *gn = sc.slot_normal + 3 * (size_t)h.slot;
          n = v3(gn[0], gn[1], gn[2]);
        }
      }
      if (P.has_plane && plane_hit(P.plane, P.plane_n, org, dir, t, n)) {
        hit = true;
        last_mat = kNoMaterial; // prim-plane.cc:34
      }
      if (P.probe && probe_on) {
        double *rec = P.probe + (size_t)(pathLength - 1) * kProbeStride;
        rec[0] = org.x; rec[1] = org.y; rec[2] = org.z; rec[3] = dir.x; rec[4] = dir.y; rec[5] = dir.z;
        rec[6] = t; rec[7] = hit ? 1.0 : 0.0; rec[8] = (h.t < kDblMax && t == h.t) ? (double)h.slot : -1.0;
        rec[9] = n.x; rec[10] = n.y; rec[11] = n.z; rec[12] = (double)last_mat; rec[13] = (double)pathLength;
        rec[14] = thr0; rec[15] = rad0;
      }
      bool path_done = false;
      if (!hit) {
        path_done = true;
        if (pathLength < 2) {
          trace_calls += 1; // eye ray -> background: radiance stays 0 (render.cc:409-412)
        } else {
          // First miss of a path that has bounced.  The reference does NOT stop: it iterates on to kMaxPathLength
          // with the stale intersection record; every one of those rays starts ~1e308 away and misses, adds
          // throughput*0.5/length and re-applies the stale material.  Their RNG draws cannot reach
          // this pixel's value, so the tail is evaluated in closed loop: same adds, same multiplies, same order.
          trace_calls += (uint32_t)P.maxPathLength;
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
      } else if (pathLength >= P.maxPathLength) {
        path_done = true;
        trace_calls += (uint32_t)P.maxPathLength;
      } else {
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
        org = hitP + scale(sd, 1.0e-3);
        dir = sd;
        ++pathLength;
      }
      if (path_done) {
        // image[...] = radiance (double -> float, render.cc:673-675), then AccumImage over passes
        acc0 += (float)rad0;
        acc1 += (float)rad1;
        acc2 += (float)rad2;
        ++pass;
        if (pass >= P.passes) {
          const size_t o = (size_t)ly * (size_t)win_w + lx;
          P.image[3 * o + 0] = acc0;
          P.image[3 * o + 1] = acc1;
          P.image[3 * o + 2] = acc2;
          if (P.count) P.count[o] += P.passes;
          state = S_NEED_PIXEL;
        } else {
          state = S_NEED_PATH;
        }
      }
    }
  }

  // ---- counters: one atomic per wave and word ---------------------------------------------------------------------
  unsigned long long v0 = trace_calls, v1 = c.rays, v2 = c.nodes, v3_ = c.tris, v4 = paths;
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
#ifdef MGPU_UTIL
    // lane 0 took part in every wave-level step it counted only while active itself, so steps are counted per lane and
    // the wave-level step count is the maximum over lanes; sums of active lanes are exact.
#endif
  }
#ifdef MGPU_UTIL
  {
    unsigned long long a = c.node_steps, b = c.tri_steps;
    for (int off = 32; off; off >>= 1) {
      a += __shfl_down(a, off);
      b += __shfl_down(b, off);
    }
    if (lane == 0) {
      atomicAdd(&P.stats[kUtilNodeSteps], a);
      atomicAdd(&P.stats[kUtilTriSteps], b);
      atomicAdd(&P.stats[kUtilOuter], (unsigned long long)u_outer);
      atomicAdd(&P.stats[kUtilTraceLanes], (unsigned long long)u_trace);
      atomicAdd(&P.stats[kUtilGenLanes], (unsigned long long)u_gen);
    }
  }
#endif
}

// =====================================================================================================================
// k_render_aov: ShowNormal / ShowUV (render.cc:458-516), the reference's two debug integrators: per pixel two draws of
// jitter, the primary ray, ONE Scene::Trace (mesh only), colour from the hit's shading normal (n * 0.5 + 0.5) or its
// texture coordinate (0.1 * s, 0, 0), black on a miss.  Primary rays of neighbouring pixels are coherent, which is the
// case the ray-per-lane traversal is best at: one lane per pixel, waves walk the frame in strides of the grid.
// =====================================================================================================================
template <int CAP>
__global__ __launch_bounds__(kBlock) void k_render_aov(DScene sc, AovParams P) {
  __shared__ __attribute__((aligned(16))) uint32_t s_stack[kBlock / 64][CAP][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t slot = (size_t)blockIdx.x * kBlock + threadIdx.x;
  Stack<CAP, true> stk;
  stk.lds = &s_stack[wave][0][lane];
  stk.overflow = sc.stack_overflow ? sc.stack_overflow + slot * sc.overflow_cap : nullptr;
  Counters c{};
  const size_t npix = (size_t)P.W * (size_t)P.H;
  for (size_t px = slot; px < npix; px += (size_t)gridDim.x * kBlock) {
    uint32_t s4[4];
    if (P.rng_mode == MGPU_RNG_TABLE) {
      const uint4 q = reinterpret_cast<const uint4 *>(P.rng_states)[px];
      s4[0] = q.x; s4[1] = q.y; s4[2] = q.z; s4[3] = q.w;
    } else {
      hash_state(P.seed, P.pass_base, (uint32_t)px, s4);
    }
    Rng rng{s4[0], s4[1], s4[2], s4[3]};
    const int gx = (int)(px % (size_t)P.W), gy = (int)(px / (size_t)P.W);
    const float ju = (float)(rng_next(rng) - 0.5);
    const float jv = (float)(rng_next(rng) - 0.5);
    const V3 org = v3(P.frame[0], P.frame[1], P.frame[2]);
    const V3 dir = camera_dir(P.frame, (double)((float)gx + ju), (double)((float)gy + jv));
    Hit h;
    traverse<CAP, true>(sc, stk, org, dir, h, c);
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    if (h.t < kDblMax) { // bvh_accel.cc:838
      if (P.mode == 0) { // BuildIntersection's shading normal (bvh_accel.cc:731-752)
        V3 n;
        if (sc.has_fv_normals) {
          const double *nn = sc.slot_normal + 9 * (size_t)h.slot;
          const double w = 1.0 - h.u - h.v;
          n = v3(w * nn[0] + h.u * nn[3] + h.v * nn[6], w * nn[1] + h.u * nn[4] + h.v * nn[7], w * nn[2] + h.u * nn[5] + h.v * nn[8]);
        } else {
          const double *gn = sc.slot_normal + 3 * (size_t)h.slot;
          n = v3(gn[0], gn[1], gn[2]);
        }
        r0 = n.x * 0.5 + 0.5;
        r1 = n.y * 0.5 + 0.5;
        r2 = n.z * 0.5 + 0.5;
      } else if (sc.fv_uvs) { // texcoord[0] (bvh_accel.cc:754-768); without uvs the reference reads an unset field: 0 here
        const double *uv = sc.fv_uvs + 6 * (size_t)sc.tris[h.slot].face;
        r0 = 0.1 * ((1.0 - h.u - h.v) * uv[0] + h.u * uv[2] + h.v * uv[4]);
      }
    }
    P.image[3 * px + 0] = (float)r0;
    P.image[3 * px + 1] = (float)r1;
    P.image[3 * px + 2] = (float)r2;
    if (P.count) P.count[px] += 1;
  }
  unsigned long long rn = c.rays, nn = c.nodes, tn = c.tris;
  for (int off = 32; off; off >>= 1) {
    rn += __shfl_down(rn, off);
    nn += __shfl_down(nn, off);
    tn += __shfl_down(tn, off);
  }
  if (lane == 0 && P.stats) {
    atomicAdd(&P.stats[kStatRays], rn);
    atomicAdd(&P.stats[kStatNodes], nn);
    atomicAdd(&P.stats[kStatTris], tn);
    atomicAdd(&P.stats[kStatTraceCalls], rn);
    atomicAdd(&P.stats[kStatPaths], rn);
  }
}

void launch_render_aov(int cap, dim3 grid, hipStream_t s, const DScene &sc, const AovParams &p) {
  switch (cap) {
  case 16: hipLaunchKernelGGL(k_render_aov<16>, grid, dim3(kBlock), 0, s, sc, p); break;
  case 24: hipLaunchKernelGGL(k_render_aov<24>, grid, dim3(kBlock), 0, s, sc, p); break;
  default: hipLaunchKernelGGL(k_render_aov<32>, grid, dim3(kBlock), 0, s, sc, p); break;
  }
}

// =====================================================================================================================
// launchers
// =====================================================================================================================
// =====================================================================================================================
// k_scene_layout: what mgpu_scene_create leaves in HBM for the traversal, one thread per BVH leaf slot.
//   tris[slot]   = p0, e1 = p1 - p0, e2 = p2 - p0 of face indices[slot] (the operands TriangleIsect forms, bvh_accel.cc:
//                  606-607), its face id and material id (kNoMaterial without a material array)
//   slot_normal  = the face's 9 face-varying normal components, or the geometric normal of BuildIntersection
//                  (bvh_accel.cc:723-729): normalize(cross(p1-p0, p2-p0)) with real3::normalize's 1e-6 guard
// fp64 subtraction, multiplication, sqrt and division are correctly rounded on the device (profiles/microbench), so the
// values equal what the host used to compute here.
// =====================================================================================================================
__global__ __launch_bounds__(256) void k_scene_layout(const double *__restrict__ verts, const uint32_t *__restrict__ faces,
                                                       const uint32_t *__restrict__ indices,
                                                       const uint32_t *__restrict__ matIDs,
                                                       const double *__restrict__ fv_normals, size_t nf,
                                                       DTri *__restrict__ tris, double *__restrict__ slot_normal) {
  const size_t slot = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (slot >= nf) return;
  const uint32_t face = indices[slot];
  const double *p0 = verts + 3 * (size_t)faces[3 * (size_t)face + 0];
  const double *p1 = verts + 3 * (size_t)faces[3 * (size_t)face + 1];
  const double *p2 = verts + 3 * (size_t)faces[3 * (size_t)face + 2];
  DTri t;
  for (int k = 0; k < 3; ++k) {
    t.p0[k] = p0[k];
    t.e1[k] = p1[k] - p0[k];
    t.e2[k] = p2[k] - p0[k];
  }
  t.face = face;
  t.mat = matIDs ? matIDs[face] : kNoMaterial;
  tris[slot] = t;
  if (fv_normals) {
    for (int k = 0; k < 9; ++k) slot_normal[9 * slot + k] = fv_normals[9 * (size_t)face + k];
  } else {
    const V3 n = normalized(cross(v3(t.e1[0], t.e1[1], t.e1[2]), v3(t.e2[0], t.e2[1], t.e2[2])));
    slot_normal[3 * slot + 0] = n.x;
    slot_normal[3 * slot + 1] = n.y;
    slot_normal[3 * slot + 2] = n.z;
  }
}

void launch_scene_layout(hipStream_t s, const double *verts, const uint32_t *faces, const uint32_t *indices,
                         const uint32_t *matIDs, const double *fv_normals, size_t nf, DTri *tris, double *slot_normal) {
  const unsigned blocks = (unsigned)((nf + 255) / 256);
  hipLaunchKernelGGL(k_scene_layout, dim3(blocks), dim3(256), 0, s, verts, faces, indices, matIDs, fv_normals, nf, tris,
                     slot_normal);
}

// =====================================================================================================================
// k_wide_layout: the WNode records of the wide traversal (mgpu_device.hpp), one thread per node index 0..nn.
// Pure copies of the reference node fields (boxes verbatim), so nothing here can change a decision.
// =====================================================================================================================
__global__ __launch_bounds__(256) void k_wide_layout(const MgpuNode *__restrict__ nodes, size_t nn, WNode *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i > nn) return;
  WNode w;
  for (int k = 0; k < 6; ++k) w.box0[k] = w.box1[k] = 0.0;
  w.ref0 = w.tag0 = w.ref1 = w.tag1 = 0u;
  for (int k = 0; k < 4; ++k) w.pad_[k] = 0u;
  auto child = [&](uint32_t c, double *box, uint32_t &ref, uint32_t &tag) {
    const MgpuNode &n = nodes[c];
    for (int k = 0; k < 3; ++k) {
      box[k] = n.bmin[k];
      box[3 + k] = n.bmax[k];
    }
    if (n.flag == 0) {
      ref = c;
      tag = kWInterior;
    } else {
      ref = n.data[1];
      tag = n.data[0]; // < kWInterior, checked by mgpu_scene_create
    }
  };
  if (i == nn) { // super root: child 0 = the root, child 1 = an empty leaf far away
    child(0u, w.box0, w.ref0, w.tag0);
    for (int k = 0; k < 6; ++k) w.box1[k] = kDblMax;
  } else if (nodes[i].flag == 0) {
    child(nodes[i].data[0], w.box0, w.ref0, w.tag0);
    child(nodes[i].data[1], w.box1, w.ref1, w.tag1);
    w.tag0 |= (uint32_t)nodes[i].axis << 30;
  }
  out[i] = w;
}

void launch_wide_layout(hipStream_t s, const MgpuNode *nodes, size_t nn, WNode *out) {
  hipLaunchKernelGGL(k_wide_layout, dim3((unsigned)((nn + 1 + 255) / 256)), dim3(256), 0, s, nodes, nn, out);
}

int pick_stack_cap(int needed_entries) {
  if (needed_entries <= 16) return 16;
  if (needed_entries <= 24) return 24;
  return 32; // deeper trees spill the remainder to the per-lane HBM overflow column
}

void launch_trace(int cap, dim3 grid, hipStream_t s, const DScene &sc, const MgpuRay *rays, size_t n,
                  MgpuIntersection *out, uint8_t *hit, unsigned long long *stats, const uint32_t *select) {
  switch (cap) {
  case 16: hipLaunchKernelGGL(k_trace<16>, grid, dim3(kBlock), 0, s, sc, rays, n, out, hit, stats, select); break;
  case 24: hipLaunchKernelGGL(k_trace<24>, grid, dim3(kBlock), 0, s, sc, rays, n, out, hit, stats, select); break;
  default: hipLaunchKernelGGL(k_trace<32>, grid, dim3(kBlock), 0, s, sc, rays, n, out, hit, stats, select); break;
  }
}

// Which batched-trace kernel suits the batch?  Rays that arrive in coherent runs (camera rays in scanline order: 64
// neighbours walk the same nodes) are traced fastest one ray per lane to completion (k_trace: 0.47 vs 0.75 ms per 4 M
// camera rays of C2); anything else by the wave-scheduled kernel (k_trace_sm: 0.64 vs 1.15 ms per 4 M random rays).
// One workgroup samples 128 groups of 64 consecutive rays spread over the batch.
__global__ __launch_bounds__(1024) void k_trace_probe(const MgpuRay *__restrict__ rays, size_t n, uint32_t *select) {
  __shared__ uint32_t coherent;
  if (threadIdx.x == 0) coherent = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t groups = n / 64; // the caller guarantees n >= 64 * 128
  uint32_t mine = 0;
  for (int k = 0; k < 8; ++k) {
    const size_t g = (size_t)(wave * 8 + k) * (groups / 128);
    const double *rd = rays[g * 64 + lane].dir;
    const double x = rd[0], y = rd[1], z = rd[2];
    const double dx = __shfl(x, 0), dy = __shfl(y, 0), dz = __shfl(z, 0);
    const double d = x * dx + y * dy + z * dz;
    const double l2 = (x * x + y * y + z * z) * (dx * dx + dy * dy + dz * dz);
    const bool near = d > 0.0 && d * d > 0.94 * l2; // cos^2 > 0.94: within ~14 degrees of the group's first direction
    if (MGPU_BALLOT(!near) == 0ull) ++mine;
  }
  if (lane == 0 && mine) atomicAdd(&coherent, mine);
  __syncthreads();
  if (threadIdx.x == 0) *select = (coherent >= 96) ? kTraceSelectV1 : kTraceSelectSm;
}

void launch_trace_probe(hipStream_t s, const MgpuRay *rays, size_t n, uint32_t *select) {
  hipLaunchKernelGGL(k_trace_probe, dim3(1), dim3(1024), 0, s, rays, n, select);
}

void launch_render(int cap, dim3 grid, hipStream_t s, const DScene &sc, const RenderParams &p) {
  switch (cap) {
  case 16: hipLaunchKernelGGL(k_render<16>, grid, dim3(kBlock), 0, s, sc, p); break;
  case 24: hipLaunchKernelGGL(k_render<24>, grid, dim3(kBlock), 0, s, sc, p); break;
  default: hipLaunchKernelGGL(k_render<32>, grid, dim3(kBlock), 0, s, sc, p); break;
  }
}

} // namespace mgpu
