// mgpu_render_env.hip -- k_render_env: RenderPanoramic (render.cc:710-763) = 10 PathTraceEnv paths (render.cc:518-590) per
// pixel over equirectangular rays (Camera::GenerateEnvRay / GenerateStereoEnvRay, camera.cc:242-329).  This is what the
// reference's console driver renders (main_console.cc:111).
//
// Same wave-scheduled traversal as k_render_sm / k_trace_sm (lane states NODE / TRI / SHADE, one body per trip of the
// wave loop), with the same two scene placements: BVH staged in LDS (1024-thread workgroup per CU) when it fits, else
// read through L1/L2 (256-thread workgroups).  What differs from PathTrace:
//   * no ground plane, no material, no throughput: a path's radiance is 0 or sum_{L = L0..maxPathLength} 0.5 / L, L0 >= 2
//     being the length at which it first missed (the reference keeps iterating after that miss with a stale record,
//     SURVEY F4; those rays start ~1e308 away and are evaluated in closed loop here, same additions in the same order);
//   * a pixel's `samples` paths draw from ONE xorshift128 stream (loop nest y, x, sample), so a lane owns a pixel for all
//     its samples, replays the draws the reference makes after a miss (3 per remaining iteration) and accumulates
//     `image[px] += radiance` exactly as written there: float += double, sample after sample;
//   * primary rays come from sin / cos of the pixel's spherical angles (device sincos: <= 1 ulp from glibc; radiance
//     depends on hit / miss events only, see DESIGN.md "Numerics").
#include <mutex>
#include <type_traits>
#include "mgpu_device.hpp"
#include "mgpu_kernels.hpp"

// deferred path start, as in k_render_sm (0 / 65: off 24.3 ms; 8 / 12: 24.1; 12 / 16: 23.6; 16 / 24: 23.9; 24 / 32: 25.5 on
// the 2048x1024 stereo panorama)
#ifndef MGPU_ENV_START_MIN
#define MGPU_ENV_START_MIN 12
#endif
#ifndef MGPU_ENV_START_FORCE
#define MGPU_ENV_START_FORCE 16
#endif
// NODE runs when cN * MGPU_ENV_NODE_WEIGHT >= cT * MGPU_ENV_TRI_WEIGHT.  Before leaves were shared between lanes NODE was
// preferred 4:1 (1: 25.9, 2: 25.8, 4: 24.2 ms on the 2048x1024 stereo panorama); with shared leaves a TRI step is cheap at a
// low lane count, and the balance is measured again below.
#ifndef MGPU_ENV_NODE_WEIGHT
#define MGPU_ENV_NODE_WEIGHT 1
#endif
#ifndef MGPU_ENV_TRI_WEIGHT
#define MGPU_ENV_TRI_WEIGHT 4
#endif

namespace mgpu {

namespace {
enum : int { ES_NODE = 0, ES_TRI = 1, ES_SHADE = 2, ES_IDLE = 3 };
constexpr int kEnvBlock = 256;
} // namespace

#ifndef MGPU_ENV_SHADE_MIN
#define MGPU_ENV_SHADE_MIN 32
#endif

template <int CAP, bool OVF, bool LDS_SCENE, int BLOCK>
__global__ __launch_bounds__(BLOCK, 4) void k_render_env(DScene sc, EnvParams P_arg) {
  MGPU_DYN_SHARED(unsigned char, smem);
  __shared__ EnvParams s_P; // launch parameters live in LDS, not in scalar registers (see k_render_sm)
  __shared__ unsigned long long s_cnt[5];
  __shared__ unsigned char s_owner[BLOCK]; // TRI step with shared leaves: lane of the k-th open leaf, per wave
  __shared__ SincosTable s_azimuth;        // the cosine sampler's azimuth table (mgpu_sincos.hpp)
  if (threadIdx.x == 0) s_P = P_arg;
  if (threadIdx.x < 5) s_cnt[threadIdx.x] = 0ull;
  sincos_table_fill(s_azimuth, threadIdx.x, BLOCK);
  __syncthreads();
  const EnvParams &P = s_P;
  uint32_t *s_stack = reinterpret_cast<uint32_t *>(smem); // [waves][CAP][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t slot = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  Stack<CAP, OVF> stk;
  stk.lds = s_stack + ((size_t)wave * CAP) * 64 + lane;
  stk.overflow = (OVF && sc.stack_overflow) ? sc.stack_overflow + slot * sc.overflow_cap : nullptr;
  // BVH in HBM: the wide traversal (mgpu_device.hpp, wide_node_step) with its far-child stack
  using WS = WStack<kWideStackLds>;
  WS wstk;
  if (!LDS_SCENE) {
    wstk.bind(smem + (size_t)wave * WS::kWaveBytes, lane);
    wstk.overflow = sc.wstack_overflow ? sc.wstack_overflow + slot * sc.woverflow_cap : nullptr;
  }
  // LDS_SCENE: nodes + triangles staged once per workgroup next to the stacks (as k_render_sm does)
  const unsigned char *lds_nodes = smem + (size_t)(BLOCK / 64) * CAP * 64 * sizeof(uint32_t);
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

  const uint32_t tiles_x = (uint32_t)(P.win_w + 7) >> 3, tiles_y = (uint32_t)(P.win_h + 7) >> 3;
  const uint32_t total_tiles = tiles_x * tiles_y;
  // wave-uniform cursor: the current 8x8 tile and the position inside it
  uint32_t cur_tile = 0, in_tile = 64;
  bool exhausted = false;

  // per-lane pixel / path state
  int st = ES_SHADE;
  bool have_ray = false, have_pixel = false;
  bool start_pending = false; // parked between two samples of its pixel: the next SHADE step it joins starts the path
  uint32_t lx = 0, ly = 0;
  int sample = 0;
  float acc = 0.0f;    // the pixel's running sum; R = G = B for this integrator
  double rad = 0.0;    // the current path's radiance
  Rng rng{1, 0, 0, 0};
  V3 org = v3(0, 0, 0), dir = v3(0, 0, 1);
  int pathLength = 1;
  // per-lane traversal state
  double ix = 0, iy = 0, iz = 0;
  uint32_t sgn = 0; // bit k: dir[k] < 0
  bool ray_plain = false; // the ray may take the min/max form of the slab test (mgpu_device.hpp, slab_hit)
  int sp = -1;           // LDS_SCENE: index of the stack top; wide form: far children on the stack
  uint32_t cur = kWNone; // wide form: record to enter next
  double bt = kDblMax, bu = 0, bv = 0;
  uint32_t bslot = kNoHit;
  uint32_t tri_cur = 0, tri_end = 0;
  uint32_t n_rays = 0, n_nodes = 0, n_tris = 0, trace_calls = 0, paths = 0;

  for (;;) {
    const unsigned long long mN = MGPU_BALLOT(st == ES_NODE);
    const unsigned long long mT = MGPU_BALLOT(st == ES_TRI);
    const unsigned long long mS = MGPU_BALLOT(st == ES_SHADE);
    const int cN = __popcll(mN), cT = __popcll(mT), cS = __popcll(mS);
    if ((cN | cT | cS) == 0) break;
    // lanes parked between paths (deferred start, see below) do not count towards the quorum; enough of them force a step
    const int cReal = __popcll(MGPU_BALLOT(st == ES_SHADE && have_ray));
    const bool run_shade = (cReal >= MGPU_ENV_SHADE_MIN) || (cN == 0 && cT == 0) || (cS - cReal >= MGPU_ENV_START_FORCE);
    if (!run_shade && cN * MGPU_ENV_NODE_WEIGHT >= cT * MGPU_ENV_TRI_WEIGHT) {
      // ================================ NODE step ================================
      const bool all_plain = MGPU_BALLOT(st == ES_NODE && !ray_plain) == 0ull; // wave-uniform
      if (st == ES_NODE) {
        const bool sx = (sgn & 1u) != 0u, sy = (sgn & 2u) != 0u, sz = (sgn & 4u) != 0u;
        // slab_hit<true> (min/max form) when every lane's ray qualifies, the literal form for this step otherwise
        auto node_pops = [&](auto plain_tag) {
          constexpr bool kPlain = decltype(plain_tag)::value;
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
            // keep the fourth load next to the other three (same 64-byte node): the compiler otherwise sinks it into the hit
            // branch as a second dependent trip -- to L1/L2 (BVH in HBM: C4 6.66 -> 6.20 ms) or to LDS (C2 6.56 -> 6.41 ms)
            MGPU_KEEP4(meta.x, meta.y, meta.z, meta.w);
            const bool hit = slab_hit<kPlain>(b0, b1, b2, org, ix, iy, iz, sx, sy, sz, bt); // IntersectRayAABB
            if (hit) {
              if (meta.x == 0) {
                const bool nearIsSecond = ((sgn >> (uint32_t)meta.y) & 1u) != 0u; // dirSign[node.axis]
                const uint32_t c0 = (uint32_t)meta.z, c1 = (uint32_t)meta.w;
                stk.put(sp + 1, nearIsSecond ? c0 : c1); // far
                stk.put(sp + 2, nearIsSecond ? c1 : c0); // near: popped first
                sp += 2;
              } else if (meta.z != 0) {
                tri_cur = (uint32_t)meta.w;
                tri_end = (uint32_t)meta.w + (uint32_t)meta.z;
                st = ES_TRI;
              }
            }
            if (st != ES_NODE || sp < 0) break;
          }
        };
        if constexpr (LDS_SCENE) {
          if (all_plain) node_pops(std::true_type{});
          else node_pops(std::false_type{});
          if (st == ES_NODE && sp < 0) st = ES_SHADE;
        } else {
          int r;
          if (all_plain)
            r = wide_node_step<true, 3, kWideStackLds>(sc.wnodes, wstk, org, ix, iy, iz, sx, sy, sz, sgn, bt, cur, sp, tri_cur, tri_end, n_nodes);
          else
            r = wide_node_step<false, 3, kWideStackLds>(sc.wnodes, wstk, org, ix, iy, iz, sx, sy, sz, sgn, bt, cur, sp, tri_cur, tri_end, n_nodes);
          if (r == WT_TRI) st = ES_TRI;
          else if (r == WT_DONE) st = ES_SHADE;
        }
      }
    } else if (!run_shade) {
      // ================================ TRI step =================================
      bool shared_done = false;
      if (cT <= 32) { // 2 or 4 lanes per open leaf (mgpu_device.hpp, shared_leaves_step)
        uint32_t my_trips = 0;
        shared_done = shared_leaves_step<LDS_SCENE, 16>(mT, cT, cT <= 16 ? 2 : 1, lane, s_owner + wave * 64, st == ES_TRI, lds_tris,
                                                        sc.tris, org, dir, tri_cur, tri_end, bt, bu, bv, bslot, n_tris, my_trips);
      }
      if (!shared_done && st == ES_TRI) {
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
      if (st == ES_TRI && tri_cur == tri_end) st = (LDS_SCENE ? sp < 0 : sp == 0) ? ES_SHADE : ES_NODE;
    } else {
      // ================================ SHADE step ===============================
      const bool shade_lane = (st == ES_SHADE);
      bool path_done = false;
      if (shade_lane) {
        path_done = start_pending || (!have_ray && !have_pixel); // parked before its next sample / asks for a pixel below
        start_pending = false;
        if (have_ray) {
          // ---- the rest of one PathTraceEnv loop iteration (render.cc:541-585) ----
          const bool hit = bt < kDblMax; // bvh_accel.cc:838
          if (!hit) {
            path_done = true;
            if (pathLength < 2) {
              trace_calls += 1; // eye ray -> background (render.cc:543-546)
            } else {
              // first miss at length L0 >= 2: the reference adds 0.5 / L for L = L0 .. maxPathLength, tracing garbage
              // rays in between and drawing three random numbers per remaining iteration (render.cc:563-574)
              trace_calls += (uint32_t)P.maxPathLength;
              if (P.maxPathLength <= 32) { // the sum from the host's table (EnvParams::tail_sum), the draws as they come
                rad += P.tail_sum[pathLength];
                for (int L = pathLength; L < P.maxPathLength; ++L) {
                  (void)rng_next(rng);
                  (void)rng_next(rng);
                  (void)rng_next(rng);
                }
              } else {
                for (int L = pathLength;; ++L) {
                  rad += 0.5 / (double)(unsigned)L;
                  if (L >= P.maxPathLength) break;
                  (void)rng_next(rng);
                  (void)rng_next(rng);
                  (void)rng_next(rng);
                }
              }
            }
          } else if (pathLength >= P.maxPathLength) {
            path_done = true;
            trace_calls += (uint32_t)P.maxPathLength;
          } else {
            V3 n;
            if (sc.has_fv_normals) { // barycentric lerp, not renormalised (bvh_accel.cc:745-748)
              const double *nn = sc.slot_normal + 9 * (size_t)bslot;
              const double w = 1.0 - bu - bv;
              n = v3(w * nn[0] + bu * nn[3] + bv * nn[6], w * nn[1] + bu * nn[4] + bv * nn[7],
                     w * nn[2] + bu * nn[5] + bv * nn[8]);
            } else {
              const double *gn = sc.slot_normal + 3 * (size_t)bslot;
              n = v3(gn[0], gn[1], gn[2]);
            }
            const V3 hitP = org + scale(dir, bt);
            (void)rng_next(rng); // `double r = randomreal();` drawn and never used (render.cc:563)
            const double ndoti = dot(n, neg(dir));
            if (ndoti < 0.0) n = neg(n);
            const V3 sd = sample_diffuse(n, rng, &s_azimuth);
            org = hitP + scale(sd, 1.0e-3);
            dir = sd;
            ++pathLength;
          }
          have_ray = false;
          if (path_done) {
            // image[...] += radiance (render.cc:749-751): float += double, one sample after the other
            acc = (float)((double)acc + rad);
            ++sample;
            if (sample >= P.samples) {
              const size_t o = (size_t)ly * (size_t)P.win_w + lx;
              P.image[3 * o + 0] = acc;
              P.image[3 * o + 1] = acc;
              P.image[3 * o + 2] = acc;
              if (P.count) P.count[o] += P.samples;
              have_pixel = false;
            }
          }
        }
      }
      // Deferred start: the prologue of a path (two sincos, the stereo rig: ~350 instructions) is worth running only for
      // many lanes at once.  With fewer than MGPU_ENV_START_MIN lanes between paths while others still traverse, those
      // lanes stay parked in SHADE without a ray and start together in a later step.
      const bool restart = shade_lane && !have_ray && (path_done || !have_pixel);
      const bool defer = (cN + cT) > 0 && __popcll(MGPU_BALLOT(restart)) < MGPU_ENV_START_MIN;
      // ---- pixel hand-out, executed by the whole wave (the cursor is wave-uniform) ----
      bool want = shade_lane && !have_pixel && !have_ray;
      for (;;) {
        const unsigned long long wm = MGPU_BALLOT(want);
        if (!wm || exhausted || defer) break;
        if (in_tile >= 64) {
          uint32_t t = 0;
          if (lane == 0) t = atomicAdd(P.work_counter, 1u);
          t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
          if (t >= total_tiles) { exhausted = true; break; }
          cur_tile = t;
          in_tile = 0;
        }
        if (want) {
          const uint32_t rank = __popcll(wm & ((1ull << lane) - 1ull));
          const uint32_t s = in_tile + rank;
          if (s < 64) {
            const uint32_t x = (cur_tile % tiles_x) * 8 + (s & 7), y = (cur_tile / tiles_x) * 8 + (s >> 3);
            if (x < (uint32_t)P.win_w && y < (uint32_t)P.win_h) { // slots of an edge tile outside the window are skipped
              lx = x; ly = y;
              have_pixel = true;
              want = false;
              // the pixel's stream (render.cc:743-753): all its samples draw from it one after the other
              const uint32_t gpix = (uint32_t)(P.y0 + (int)y) * (uint32_t)P.W + (uint32_t)(P.x0 + (int)x);
              uint32_t s4[4];
              if (P.rng_mode == MGPU_RNG_TABLE) {
                const uint4 q = reinterpret_cast<const uint4 *>(P.rng_states)[gpix];
                s4[0] = q.x; s4[1] = q.y; s4[2] = q.z; s4[3] = q.w;
              } else {
                hash_state(P.seed, P.pass_base, gpix, s4);
              }
              rng = Rng{s4[0], s4[1], s4[2], s4[3]};
              acc = 0.0f; // memset(image) of render.cc:737
              sample = 0;
              path_done = true;
            }
          }
        }
        in_tile += (uint32_t)__popcll(wm);
        if (in_tile > 64) in_tile = 64;
      }
      // ---- next path / next traversal ----
      if (shade_lane) {
        if (restart && defer) {
          start_pending = have_pixel; // stays in SHADE without a ray; resumes with the prologue (or asks for a pixel)
        } else if (!have_pixel) {
          st = ES_IDLE; // no pixel left for this lane
        } else {
          if (path_done) {
            // PathTraceEnv prologue (render.cc:523-534): jitter, equirectangular ray
            const float ju = (float)(rng_next(rng) - 0.5);
            const float jv = (float)(rng_next(rng) - 0.5);
            const double u = (double)((float)(P.x0 + (int)lx) + ju), v = (double)((float)(P.y0 + (int)ly) + jv);
            const double kPi = 3.14159265358979323846, k2Pi = 6.283185307179586; // M_PI, 2.0 * M_PI
            const double phi = k2Pi * (u / (double)P.W);
            double sin_p, cos_p, sin_t, cos_t;
            sincos(phi, &sin_p, &cos_p);
            if (!P.stereo) { // Camera::GenerateEnvRay, camera.cc:242-257
              const double theta = kPi * (v / (double)P.H);
              sincos(theta, &sin_t, &cos_t);
              org = v3(P.origin[0], P.origin[1], P.origin[2]);
              dir = v3(sin_t * cos_p, cos_t, sin_t * sin_p);
            } else { // Camera::GenerateStereoEnvRay, camera.cc:259-329
              const bool left = v < (double)(P.H >> 1);
              const double theta = kPi * fmod(2.0 * v / (double)P.H, 1.0);
              sincos(theta, &sin_t, &cos_t);
              const V3 d0 = v3(sin_t * cos_p, cos_t, sin_t * sin_p);
              V3 par = left ? v3(-d0.z, 0.0, d0.x) : v3(d0.z, 0.0, -d0.x);
              par = scale(normalized_w(par), 0.5);
              org = v3(P.origin[0] + par.x, P.origin[1] + par.y, P.origin[2] + par.z);
              // psi = atan2(0.5, 4), negated for the left eye; cos / sin of it come from the host's libm
              const double cpsi = P.cos_psi, spsi = left ? -P.sin_psi : P.sin_psi;
              dir = normalized_w(v3(d0.x * cpsi - d0.z * spsi, d0.y, d0.x * spsi + d0.z * cpsi));
            }
            rad = 0.0;
            pathLength = 1;
            ++paths;
          }
          // arm the traversal of (org, dir): BVHAccel::Traverse prologue, bvh_accel.cc:774-802
          sgn = (dir.x < 0.0 ? 1u : 0u) | (dir.y < 0.0 ? 2u : 0u) | (dir.z < 0.0 ? 4u : 0u);
          const bool inv_ok = inverse_dir_w(dir, ix, iy, iz); // 1.0 / dir, no zero guard, as the reference
          ray_plain = sc.boxes_ordered && inv_ok && origin_is_finite(org);
          bt = kDblMax; bu = 0.0; bv = 0.0; bslot = kNoHit;
          sp = 0;
          if constexpr (LDS_SCENE) {
            stk.put(0, 0u);
          } else {
            cur = sc.wroot; // the super root (child 0 = the tree's root)
            n_nodes -= 1u;
          }
          have_ray = true;
          ++n_rays;
          st = ES_NODE;
        }
      }
    }
  }

  // counters: wave reduction, one LDS atomic per wave, one global atomic per workgroup and word
  unsigned long long v0 = trace_calls, v1 = n_rays, v2 = n_nodes, v3_ = n_tris, v4 = paths;
  for (int off = 32; off; off >>= 1) {
    v0 += __shfl_down(v0, off);
    v1 += __shfl_down(v1, off);
    v2 += __shfl_down(v2, off);
    v3_ += __shfl_down(v3_, off);
    v4 += __shfl_down(v4, off);
  }
  if (lane == 0) {
    atomicAdd(&s_cnt[0], v0);
    atomicAdd(&s_cnt[1], v1);
    atomicAdd(&s_cnt[2], v2);
    atomicAdd(&s_cnt[3], v3_);
    atomicAdd(&s_cnt[4], v4);
  }
  __syncthreads();
  if (threadIdx.x == 0 && P.stats) {
    atomicAdd(&P.stats[kStatTraceCalls], s_cnt[0]);
    atomicAdd(&P.stats[kStatRays], s_cnt[1]);
    atomicAdd(&P.stats[kStatNodes], s_cnt[2]);
    atomicAdd(&P.stats[kStatTris], s_cnt[3]);
    atomicAdd(&P.stats[kStatPaths], s_cnt[4]);
  }
}

template <int CAP, bool OVF, bool LDS_SCENE, int BLOCK>
static hipError_t launch_one(dim3 grid, hipStream_t s, const DScene &sc, const EnvParams &p) {
  size_t shmem = LDS_SCENE ? (size_t)(BLOCK / 64) * CAP * 64 * sizeof(uint32_t) : (size_t)(BLOCK / 64) * WStack<kWideStackLds>::kWaveBytes;
  if (LDS_SCENE) {
    shmem += (size_t)p.lds_nodes_bytes + (size_t)p.lds_tris_bytes;
    // the attribute is per function and device (dynamic + static <= 160 KB); set it whenever a launch needs more than any
    // before it on that device
    static size_t granted[16] = {0};
    static std::mutex granted_mutex;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(granted_mutex);
    if (dev < 0 || dev >= 16 || shmem > granted[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_render_env<CAP, OVF, LDS_SCENE, BLOCK>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) granted[dev] = shmem;
    }
  }
  hipLaunchKernelGGL((k_render_env<CAP, OVF, LDS_SCENE, BLOCK>), grid, dim3(BLOCK), shmem, s, sc, p);
  return hipGetLastError();
}

hipError_t launch_render_env(int cap, bool lds_scene, dim3 grid, hipStream_t s, const DScene &sc, const EnvParams &p) {
  const bool ovf = sc.overflow_cap != 0;
  if (lds_scene && !ovf) {
    if (cap == 16) return launch_one<16, false, true, 1024>(grid, s, sc, p);
    if (cap == 24) return launch_one<24, false, true, 1024>(grid, s, sc, p);
    return hipErrorInvalidConfiguration;
  }
  return launch_one<1, false, false, kEnvBlock>(grid, s, sc, p); // BVH in HBM: the wide form, one variant
}

} // namespace mgpu
