// mgpu_render_w5.hip -- k_render_w5: the wave-scheduled path tracer for HBM-resident scenes with its per-lane state divided BY HAND
// between registers and LDS, so that five waves per SIMD (20 per CU, <= 96 VGPRs) hold it without the allocator spilling across
// the traversal bodies.
//
// Same state machine, same per-ray operation order and same counters as k_render_sm<.., LDS_SCENE = false, ..> (mgpu_render_sm.hip,
// whose header describes the three bodies and the work hand-out); what differs is where a lane's state lives:
//
//   registers (live across every body)   org, dir, 1/dir, best t / u / v / slot, state, stack depth, record or triangle run under
//                                        way, direction signs + "plain ray" + "probe" bits in one word, node / triangle counters
//   LDS, lane-minor, SHADE only (kCold)  the xorshift128 state (4), throughput (2; grey scenes: one channel), pixel (x | y << 16),
//                                        pass | pathLength << 16 | "was handed a path" << 24, last material, the path's cost base
//                                        -- loaded where SHADE begins, stored where it ends, never live in NODE or TRI
//   LDS, per wave                        rays / Trace() calls / paths counters (booked from ballots by one lane per SHADE step)
//   LDS, lane-minor                      the far-child stack with 12-byte entries (mgpu_device.hpp, WStackP): K entries, deeper
//                                        ones in the lane's HBM column
//   folded away                          `have_ray` is a state of its own (W_SHADE = a ray to finish, W_PARK = between paths)
//
// Grey scenes only (every material the reference loads from .obj / .eson: three equal channels), references that fit WStackP's
// packing, pathLength <= 255, windows up to 65 535 pixels a side: anything else takes k_render_sm.  (bvh_accel.cc:773-844 the walk,
// render.cc:381-456 the path, 657-681 the pixel loop.)
#include "mgpu_device.hpp"
#include "mgpu_kernels.hpp"

#include <mutex>

namespace mgpu {

enum : int { W_NODE = 0, W_TRI = 1, W_SHADE = 2, W_IDLE = 3, W_PARK = 4 };

#ifndef MGPU_W5_STACK
#define MGPU_W5_STACK 5 // far-child stack entries per lane in LDS (12 bytes each)
#endif
#ifndef MGPU_W5_START_MIN
#define MGPU_W5_START_MIN 8
#define MGPU_W5_START_FORCE 12
#endif
#ifndef MGPU_W5_SHADE_MIN
#define MGPU_W5_SHADE_MIN 36
#endif
#ifndef MGPU_W5_TRI_WEIGHT
#define MGPU_W5_TRI_WEIGHT 4
#endif
#ifndef MGPU_W5_SHARE4_MAX
#define MGPU_W5_SHARE4_MAX 16
#endif
#ifndef MGPU_W5_TRIS_PER_STEP
#define MGPU_W5_TRIS_PER_STEP 8
#endif
#ifndef MGPU_W5_WIDE_PER_STEP
#define MGPU_W5_WIDE_PER_STEP 3
#endif
#ifndef MGPU_W5_SHARED_LEAVES
#define MGPU_W5_SHARED_LEAVES 1
#endif
#ifndef MGPU_W5_WAVES
#define MGPU_W5_WAVES 5
#endif

constexpr int kW5Stack = MGPU_W5_STACK;
constexpr int kColdWords = 10;                   // dwords of SHADE-only state per lane
constexpr uint32_t kFlagPlain = 8u, kFlagProbe = 16u; // beside the three direction-sign bits

int render_w5_stack_entries() { return kW5Stack; }
size_t render_w5_wave_bytes() { return WStackP<kW5Stack>::kWaveBytes + (size_t)kColdWords * 64 * 4; }

template <int BLOCK>
__global__ __launch_bounds__(BLOCK, MGPU_W5_WAVES) void k_render_w5(DScene sc, RenderParams P_arg) {
  __shared__ RenderParams s_P;
  __shared__ SincosTable s_azimuth;
  if (threadIdx.x == 0) s_P = P_arg;
  sincos_table_fill(s_azimuth, threadIdx.x, BLOCK);
  __syncthreads();
  const RenderParams &P = s_P;
  MGPU_DYN_SHARED(unsigned char, smem);
  constexpr int kWaves = BLOCK / 64;
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  using WS = WStackP<kW5Stack>;
  WS wstk;
  wstk.bind(smem + (size_t)wave * WS::kWaveBytes, lane);
  wstk.overflow = sc.wstack_overflow ? sc.wstack_overflow + gid * sc.woverflow_cap : nullptr;
  // SHADE-only state: word k of this lane at cold[k * 64]
  lds_u32 *cold = (lds_u32 *)(smem + (size_t)kWaves * WS::kWaveBytes) + (size_t)wave * kColdWords * 64 + lane;
  // the treelet (mgpu_device.hpp, kWTreelet) behind both
  constexpr bool TL = true;
  const unsigned char *lds_treelet = smem + (size_t)kWaves * (WS::kWaveBytes + (size_t)kColdWords * 64 * 4);
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(sc.treelet);
    uint4 *dst = reinterpret_cast<uint4 *>(const_cast<unsigned char *>(lds_treelet));
    const uint32_t n16 = P.lds_nodes_bytes >> 4;
    for (uint32_t i = threadIdx.x; i < n16; i += BLOCK) dst[i] = src[i];
  }
  const int win_w = P.x1 - P.x0;
  const uint32_t tiles_x = (uint32_t)(win_w + 7) >> 3;
  const uint32_t tiles_y = (uint32_t)(P.n_rows + 7) >> 3;
  const uint32_t total_tiles = tiles_x * tiles_y;
  const uint32_t total_items = total_tiles * (uint32_t)P.passes;
  uint32_t in_item = 64;
  bool exhausted = false;
  constexpr uint32_t kWgChunk = BLOCK >= 512 ? 12u : 8u;
  const uint32_t shard_items = (total_items + (uint32_t)kShards - 1) / (uint32_t)kShards;
  uint32_t home_shard = 0;
  uint32_t item_tile = 0, item_pass = 0;
  MGPU_XCC_ID(home_shard);
  home_shard &= 7u;
  __shared__ unsigned char s_owner[BLOCK];
  __shared__ unsigned long long wg_cursor;
  __shared__ uint32_t wg_lock, wg_shard_off, wg_dry;
  __shared__ uint32_t s_wcnt[kWaves][4]; // per wave: rays, Trace() calls, paths
  if (threadIdx.x == 0) {
    wg_cursor = 0ull;
    wg_lock = 0u;
    wg_shard_off = 0u;
    wg_dry = 0u;
  }
  if (lane < 4) s_wcnt[wave][lane] = 0u;
  for (int k = 0; k < kColdWords; ++k) cold[k * 64] = 0u;
  __syncthreads();

  // ---- registers: what NODE and TRI need --------------------------------------------------------------------------------------
  int st = W_PARK; // everybody starts by asking for work
  V3 org = v3(0, 0, 0), dir = v3(0, 0, 1);
  double ix = 0, iy = 0, iz = 0;
  uint32_t flags = 0; // bits 0..2: dir[k] < 0 (dirSign, bvh_accel.cc:786-790); kFlagPlain; kFlagProbe
  int sp = 0;         // far children on the stack
  uint32_t cur = kWNone;
  double bt = kDblMax, bu = 0, bv = 0;
  uint32_t bslot = kNoHit;
  uint32_t tri_cur = 0, tri_end = 0;
  uint32_t n_nodes = 0, n_tris = 0;

  for (;;) {
    const unsigned long long mN = MGPU_BALLOT(st == W_NODE);
    const unsigned long long mT = MGPU_BALLOT(st == W_TRI);
    const unsigned long long mR = MGPU_BALLOT(st == W_SHADE);
    const unsigned long long mP = MGPU_BALLOT(st == W_PARK);
    const int cN = __popcll(mN), cT = __popcll(mT), cReal = __popcll(mR), cPark = __popcll(mP);
    if ((cN | cT | cReal | cPark) == 0) break;
    // the rule of k_render_sm: SHADE with a quorum of rays to finish, or of lanes between paths, or when nothing else can run
    const bool run_shade = (cReal >= MGPU_W5_SHADE_MIN) || (cN == 0 && cT == 0) || (cPark >= MGPU_W5_START_FORCE);
    if (!run_shade && cN >= cT * MGPU_W5_TRI_WEIGHT) {
      // ================================ NODE step ================================
      const bool all_plain = MGPU_BALLOT(st == W_NODE && (flags & kFlagPlain) == 0u) == 0ull;
      if (st == W_NODE) {
        const bool sx = (flags & 1u) != 0u, sy = (flags & 2u) != 0u, sz = (flags & 4u) != 0u;
        int r;
        if (all_plain)
          r = wide_node_step<true, MGPU_W5_WIDE_PER_STEP, kW5Stack, TL, WS>(sc.wnodes, wstk, org, ix, iy, iz, sx, sy, sz, flags & 7u, bt, cur, sp, tri_cur,
                                                                           tri_end, n_nodes, lds_treelet);
        else
          r = wide_node_step<false, MGPU_W5_WIDE_PER_STEP, kW5Stack, TL, WS>(sc.wnodes, wstk, org, ix, iy, iz, sx, sy, sz, flags & 7u, bt, cur, sp, tri_cur,
                                                                            tri_end, n_nodes, lds_treelet);
        if (r == WT_TRI) st = W_TRI;
        else if (r == WT_DONE) st = W_SHADE;
      }
    } else if (!run_shade) {
      // ================================ TRI step =================================
      bool shared_done = false;
#if MGPU_W5_SHARED_LEAVES
      if (cT <= 32) {
        uint32_t my_trips = 0;
        shared_done = shared_leaves_step<false, MGPU_W5_TRIS_PER_STEP>(mT, cT, cT <= MGPU_W5_SHARE4_MAX ? 2 : 1, lane, s_owner + wave * 64, st == W_TRI, nullptr,
                                                                      sc.tris, org, dir, tri_cur, tri_end, bt, bu, bv, bslot, n_tris, my_trips);
      }
#endif
      if (!shared_done && st == W_TRI) {
#pragma unroll 1
        for (int rep = 0; rep < MGPU_W5_TRIS_PER_STEP; ++rep) {
          const DTri *tp = sc.tris + tri_cur;
          const double2 a0 = reinterpret_cast<const double2 *>(tp)[0], a1 = reinterpret_cast<const double2 *>(tp)[1],
                        a2 = reinterpret_cast<const double2 *>(tp)[2], a3 = reinterpret_cast<const double2 *>(tp)[3];
          const double e2z = tp->e2[2];
          // TriangleIsect, bvh_accel.cc:595-638
          ++n_tris;
          const V3 p0 = v3(a0.x, a0.y, a1.x), e1 = v3(a1.y, a2.x, a2.y), e2 = v3(a3.x, a3.y, e2z);
          const V3 p = cross(dir, e2);
          const double det = dot(e1, p);
          if (!(fabs(det) < kDblEps1024)) {
            const double invDet = inv_det_w(det);
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
      if (st == W_TRI && tri_cur == tri_end) st = sp == 0 ? W_SHADE : W_NODE;
    } else {
      // ================================ SHADE step ===============================
      // (1) lanes with a ray finish it; (2) the whole wave runs the path hand-out; (3) lanes start their next path / arm their
      // next traversal.  The cold state is loaded here and stored at the end: it is live in this body only.
      const bool shade_lane = st == W_SHADE || st == W_PARK;
      const bool have_ray = st == W_SHADE;
      Rng rng{0, 0, 0, 0};
      double thr0 = 0.0;
      uint32_t lx = 0, ly = 0, pass = 0, last_mat = 0, cost_base = 0;
      int pathLength = 0;
      bool have_path = false;
      if (shade_lane) {
        rng = Rng{cold[0], cold[64], cold[128], cold[192]};
        thr0 = __hiloint2double((int)cold[5 * 64], (int)cold[4 * 64]);
        const uint32_t xy = cold[6 * 64], pp = cold[7 * 64];
        lx = xy & 0xFFFFu;
        ly = xy >> 16;
        pass = pp & 0xFFFFu;
        pathLength = (int)((pp >> 16) & 0xFFu);
        have_path = (pp >> 24) != 0u;
        last_mat = cold[8 * 64];
        cost_base = cold[9 * 64];
      }
      bool path_done = false, want_pixel = false;
      uint32_t tc_add = 0; // Scene::Trace calls the reference makes for what ends here
      if (shade_lane) {
        path_done = !have_ray;
        if (have_ray) {
          // ---- the rest of one PathTrace loop iteration (render.cc:403-452) ----
          bool hit = bt < kDblMax; // bvh_accel.cc:838
          double t = bt;
          V3 n = v3(0, 0, 0);
          if (bslot != kNoHit) last_mat = sc.tris[bslot].mat; // written by TestLeafNode on every accepted triangle
          if (hit) {
            if (sc.has_fv_normals) { // barycentric lerp, not renormalised (bvh_accel.cc:745-748)
              const double *nn = sc.slot_normal + 9 * (size_t)bslot;
              const double w = 1.0 - bu - bv;
              n.x = w * nn[0] + bu * nn[3] + bv * nn[6];
              n.y = w * nn[1] + bu * nn[4] + bv * nn[7];
              n.z = w * nn[2] + bu * nn[5] + bv * nn[8];
            } else {
              const double *gn = sc.slot_normal + 3 * (size_t)bslot;
              n = v3(gn[0], gn[1], gn[2]);
            }
          }
          if (P.has_plane && plane_hit(P.plane, P.plane_n, org, dir, t, n)) {
            hit = true;
            last_mat = kNoMaterial; // prim-plane.cc:34
          }
          if (P.probe && (flags & kFlagProbe) != 0u) {
            double *rec = P.probe + (size_t)(pathLength - 1) * kProbeStride;
            rec[0] = org.x; rec[1] = org.y; rec[2] = org.z; rec[3] = dir.x; rec[4] = dir.y; rec[5] = dir.z;
            rec[6] = t; rec[7] = hit ? 1.0 : 0.0; rec[8] = (bt < kDblMax && t == bt) ? (double)bslot : -1.0;
            rec[9] = n.x; rec[10] = n.y; rec[11] = n.z; rec[12] = (double)last_mat; rec[13] = (double)pathLength;
            rec[14] = thr0; rec[15] = 0.0;
          }
          double rad0 = 0.0;
          if (!hit) {
            path_done = true;
            if (pathLength < 2) {
              tc_add = 1u; // eye ray -> background: radiance stays 0 (render.cc:409-412)
            } else {
              // first miss of a path that has bounced: the reference's stale-record tail in closed loop (k_render_sm)
              tc_add = (uint32_t)P.maxPathLength;
              double d0 = 0.5; // Material().diffuse default (material.h:12-15)
              const bool mul = last_mat != kNoMaterial;
              if (mul && (size_t)(int)last_mat < (size_t)sc.nm) d0 = sc.mat_diffuse[3 * (size_t)last_mat + 0];
              const unsigned long long thr_bits = (unsigned long long)__double_as_longlong(thr0);
              const bool unit_ok = P.maxPathLength <= 16 && d0 == 0.5 && (thr_bits & 0x000FFFFFFFFFFFFFull) == 0ull && thr0 >= 0x1p-900 && thr0 <= 1.0;
              if (!MGPU_ANY(!unit_ok)) { // (both forms give the same bits: the table when every lane's operands qualify)
                rad0 = thr0 * P.tail_unit[mul ? 1 : 0][pathLength];
              } else if (P.maxPathLength <= 16) { // x / L through the rounded reciprocal (mgpu_kernels.hpp, inv_len)
                for (int L = pathLength;; ++L) {
                  const double x = thr0 * 0.5, y = P.inv_len[L], dl = (double)(unsigned)L;
                  const double q = x * y;
                  rad0 += fma(fma(-q, dl, x), y, q);
                  if (L >= P.maxPathLength) break;
                  if (mul) thr0 *= d0;
                }
              } else {
                for (int L = pathLength;; ++L) {
                  rad0 += thr0 * 0.5 / (double)(unsigned)L;
                  if (L >= P.maxPathLength) break;
                  if (mul) thr0 *= d0;
                }
              }
            }
          } else if (pathLength >= P.maxPathLength) {
            path_done = true;
            tc_add = (uint32_t)P.maxPathLength;
          } else {
            const V3 hitP = org + scale(dir, t);
            (void)rng_next(rng); // `double r = randomreal();` drawn and never used (render.cc:430)
            const double ndoti = dot(n, neg(dir));
            if (ndoti < 0.0) n = neg(n);
            const V3 sd = sample_diffuse_t<true>(n, rng, &s_azimuth);
            if (last_mat != kNoMaterial) { // Scene::GetMaterial, scene.h:58-65
              if ((size_t)(int)last_mat < (size_t)sc.nm) thr0 *= sc.mat_diffuse[3 * (size_t)last_mat + 0];
              else thr0 *= 0.5;
            }
            org = hitP + scale(sd, 1.0e-3);
            dir = sd;
            ++pathLength;
          }
          if (path_done) {
            // grey: one float per pixel and pass (k_accumulate_tiled_mono), or the image's three channels when there is one pass
            if (P.pass_stride) {
              P.out[(size_t)pass * P.pass_stride + ((size_t)(ly >> 3) * tiles_x + (lx >> 3)) * 64u + (size_t)(((ly & 7u) << 3) + (lx & 7u))] = (float)rad0;
            } else {
              float *dst = P.out + 3 * ((size_t)ly * (size_t)win_w + lx);
              dst[0] = (float)rad0;
              dst[1] = (float)rad0;
              dst[2] = (float)rad0;
            }
            if (P.tile_cost && pass == 0u) // what this path cost (its rays = its length), for the next launch's hand-out order
              atomicAdd(P.tile_cost + ((ly >> 3) * tiles_x + (lx >> 3)), n_nodes + n_tris - cost_base + 16u * (uint32_t)pathLength);
          }
        }
        want_pixel = path_done;
      }

      // ---- (2) path hand-out, executed by the whole wave (cursor variables are wave-uniform): k_render_sm's, item for item ----
      const bool defer = !exhausted && (cN + cT) > 0 && __popcll(MGPU_BALLOT(want_pixel)) < MGPU_W5_START_MIN;
      for (;;) {
        const unsigned long long want = MGPU_BALLOT(want_pixel);
        if (!want || exhausted || defer) break;
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
            if (__builtin_amdgcn_readfirstlane((int)flag)) {
              exhausted = true;
              break;
            }
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
                  const uint32_t n_sh = sh * shard_items < total_items ? min(shard_items, total_items - sh * shard_items) : 0u;
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
          const uint32_t item = cur_shard * shard_items + item_local;
          const uint32_t ti = item / (uint32_t)P.passes;
          item_pass = item - ti * (uint32_t)P.passes;
          item_tile = P.tile_order ? (uint32_t)__builtin_amdgcn_readfirstlane((int)P.tile_order[ti]) : ti;
        }
        if (want_pixel) {
          const uint32_t rank = __popcll(want & ((1ull << lane) - 1ull));
          const uint32_t slot = in_item + rank;
          if (slot < 64) {
            const uint32_t tx = item_tile % tiles_x, ty = item_tile / tiles_x;
            const uint32_t x = tx * 8 + (slot & 7), y = ty * 8 + (slot >> 3);
            if (x < (uint32_t)win_w && y < (uint32_t)P.n_rows) {
              lx = x;
              ly = y;
              pass = item_pass;
              have_path = true;
              want_pixel = false;
            }
          }
        }
        in_item += (uint32_t)__popcll(want);
        if (in_item >= 64) in_item = 64;
      }

      // ---- (3) next path / next traversal ----
      bool started = false, armed = false;
      if (shade_lane) {
        if (path_done && have_path) {
          have_path = false;
          started = true;
          // a new eye path (PathTrace prologue, render.cc:387-400)
          const int gy = (P.y_first + (int)(ly / (uint32_t)P.strip_h) * P.y_period + (int)(ly % (uint32_t)P.strip_h)) * P.pix_step;
          const int gx = (P.x0 + (int)lx) * P.pix_step;
          const uint32_t gpix = (uint32_t)gy * (uint32_t)P.W + (uint32_t)gx;
          uint32_t s4[4];
          if (P.rng_mode == MGPU_RNG_TABLE) {
            const uint4 q = reinterpret_cast<const uint4 *>(P.rng_states)[(size_t)pass * P.W * P.H + gpix];
            s4[0] = q.x; s4[1] = q.y; s4[2] = q.z; s4[3] = q.w;
          } else {
            hash_state(P.seed, P.pass_base + pass, gpix, s4);
          }
          rng = Rng{s4[0], s4[1], s4[2], s4[3]};
          flags = (P.probe && gpix == P.probe_pixel && pass == P.probe_pass) ? kFlagProbe : 0u;
          const float ju = (float)(rng_next(rng) - 0.5);
          const float jv = (float)(rng_next(rng) - 0.5);
          org = v3(P.frame[0], P.frame[1], P.frame[2]);
          dir = camera_dir(P.frame, (double)((float)gx + ju), (double)((float)gy + jv));
          thr0 = 1.0;
          pathLength = 1;
          cost_base = n_nodes + n_tris;
          path_done = false;
        }
        if (path_done) {
          st = exhausted ? W_IDLE : W_PARK;
        } else {
          // arm the traversal of (org, dir): BVHAccel::Traverse prologue, bvh_accel.cc:774-802
          const uint32_t sgn = (dir.x < 0.0 ? 1u : 0u) | (dir.y < 0.0 ? 2u : 0u) | (dir.z < 0.0 ? 4u : 0u);
          const bool inv_ok = inverse_dir_w(dir, ix, iy, iz); // 1.0 / dir, no zero guard, as the reference
          const bool plain = sc.boxes_ordered && inv_ok && origin_is_finite(org);
          flags = (flags & kFlagProbe) | sgn | (plain ? kFlagPlain : 0u);
          bt = kDblMax; bu = 0.0; bv = 0.0; bslot = kNoHit;
          sp = 0;
          armed = true;
          st = W_NODE;
          cur = kWTreelet; // the super root (record 0 of the treelet): its child 0 is the tree's root (the reference's first pop)
          n_nodes -= 1u;   // ... and its child 1 a dummy the reference never pops
        }
        // the cold state goes back
        cold[0] = rng.x; cold[64] = rng.y; cold[128] = rng.z; cold[192] = rng.w;
        cold[4 * 64] = (uint32_t)__double2loint(thr0);
        cold[5 * 64] = (uint32_t)__double2hiint(thr0);
        cold[6 * 64] = lx | (ly << 16);
        cold[7 * 64] = pass | ((uint32_t)pathLength << 16) | (have_path ? 1u << 24 : 0u);
        cold[8 * 64] = last_mat;
        cold[9 * 64] = cost_base;
      }
      // the wave's counters: rays armed, paths started, Trace() calls of what ended (0, 1 or maxPathLength per lane)
      {
        const uint32_t a = (uint32_t)__popcll(MGPU_BALLOT(armed)), s = (uint32_t)__popcll(MGPU_BALLOT(started));
        const uint32_t t1 = (uint32_t)__popcll(MGPU_BALLOT(tc_add == 1u)), tm = (uint32_t)__popcll(MGPU_BALLOT(tc_add > 1u));
        if (lane == 0) {
          s_wcnt[wave][0] += a;
          s_wcnt[wave][1] += t1 + tm * (uint32_t)P.maxPathLength;
          s_wcnt[wave][2] += s;
        }
      }
    }
  }

  // ---- counters: one atomic per wave and word -----------------------------------------------------------------
  unsigned long long v2 = n_nodes, v3_ = n_tris;
  for (int off = 32; off; off >>= 1) {
    v2 += __shfl_down(v2, off);
    v3_ += __shfl_down(v3_, off);
  }
  if (lane == 0 && P.stats) {
    atomicAdd(&P.stats[kStatTraceCalls], (unsigned long long)s_wcnt[wave][1]);
    atomicAdd(&P.stats[kStatRays], (unsigned long long)s_wcnt[wave][0]);
    atomicAdd(&P.stats[kStatNodes], v2);
    atomicAdd(&P.stats[kStatTris], v3_);
    atomicAdd(&P.stats[kStatPaths], (unsigned long long)s_wcnt[wave][2]);
  }
}

template <int BLOCK> static hipError_t launch_w5(dim3 grid, hipStream_t s, size_t shmem, const DScene &sc, const RenderParams &p) {
  auto kern = k_render_w5<BLOCK>;
  static size_t granted[16] = {0};
  static std::mutex granted_mutex;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (shmem > 48 * 1024) {
    std::lock_guard<std::mutex> lock(granted_mutex);
    if (dev < 0 || dev >= 16 || shmem > granted[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) granted[dev] = shmem;
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(BLOCK), shmem, s, sc, p);
  return hipGetLastError();
}

// block: 640 (two workgroups of ten waves per CU) or 320 (four of five); shmem = block / 64 * render_w5_wave_bytes() + the treelet
hipError_t launch_render_w5(int block, dim3 grid, hipStream_t s, size_t shmem, const DScene &sc, const RenderParams &p) {
  if (block == 640) return launch_w5<640>(grid, s, shmem, sc, p);
  if (block == 320) return launch_w5<320>(grid, s, shmem, sc, p);
  return hipErrorInvalidConfiguration;
}

} // namespace mgpu
