// mgpu_render_f32.hip -- k_render_f32: the FAST MODE of the renderer.  NOT the product's default and not
// bit-exact: the same algorithm as k_render_sm -- PathTrace / BVHAccel::Traverse with the reference's visiting order, leaf
// order, random draws and post-miss continuation -- evaluated in float instead of double:
//   * nodes are 32 bytes (box in float, rounded OUTWARD and widened by a few ulp so that rounding never opens a crack
//     between a box and what it holds; children / leaf run packed in 8 bytes), triangles 48 bytes (p0, e1, e2 in float +
//     material), shading normals and materials float copies of the scene's arrays (k_layout_f32, once per scene);
//   * the traversal, Moeller-Trumbore, the cosine sample and the throughput / radiance arithmetic run in fp32 (the
//     hardware's v_rsq / v_rcp / v_sin / v_cos for normalisation, 1 / dir and the azimuth: ~1e-6 relative);
//   * the random stream is the reference's xorshift128 with the per-(pixel, pass) hash seeding of MGPU_RNG_HASH, a draw
//     being its top 24 bits.
// A path therefore follows the reference's path until a hit / miss decision falls differently (a ray within ~1e-6 of a
// silhouette), after which that path is a different sample of the same integrand.  tests/test_gpu_parity.py measures the
// distance to the fp64 frame (per-pixel L2: rms and the share of pixels that moved) and DESIGN.md 5 reports it next to
// north_star's 1e-4; bench.py reports the mode as an extra line, never as `value`.
//
// Structure (states NODE / TRI / SHADE, one body per trip, work items = (8x8 tile, pass), per-XCD cursors, cost-ordered
// hand-out, shared leaves, deferred path start) is k_render_sm's -- see mgpu_render_sm.hip for the why of every rule; what
// is absent here is everything that exists there for exactness or diagnosis (probe, occupancy accounting, the wide records
// with their exact tmin re-test, the NaN-faithful triangle loop, the GREY instantiation).
#include <mutex>

#include "mgpu_device.hpp"
#include "mgpu_kernels.hpp"

// Waves per SIMD of the HBM-resident variant: the float state needs ~96 VGPRs, so five fit where the fp64 kernel runs four
// (1 M-triangle grid 5.00 -> 4.49 ms, teapot 4.27 -> 3.88 per 16 spp; six spill: 4.69 / 4.44) -- as long as the stacks of 20
// waves fit in LDS, i.e. up to 24 entries per lane.
#ifndef MGPU_F32_HBM_WAVES
#define MGPU_F32_HBM_WAVES(cap) ((cap) <= 24 ? 5 : 4)
#endif

namespace mgpu {

namespace {

enum : int { F_NODE = 0, F_TRI = 1, F_SHADE = 2, F_IDLE = 3 };
constexpr int kNodesPerStep = 6, kTrisPerStep = 16, kShadeMin = 36;
constexpr uint32_t kLeafTag = 3u; // FNode::b >> 30

struct F3 {
  float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 fscale(F3 a, float f) { return f3(a.x * f, a.y * f, a.z * f); }
__device__ __forceinline__ F3 fneg(F3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ F3 fcross(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float fdot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// real3::normalize (common.h:48-56): vectors shorter than 1e-6 stay as they are
__device__ __forceinline__ F3 fnormalized(F3 a) {
  const float l2 = fdot(a, a);
  return l2 > 1.0e-12f ? fscale(a, __builtin_amdgcn_rsqf(l2)) : a;
}
// randomreal() (render.cc:137-168): the same state update; the draw is the top 24 bits (an fp32 in [0, 1))
__device__ __forceinline__ float rng_next_f(Rng &r) {
  const uint32_t t = r.x ^ (r.x << 11);
  r.x = r.y;
  r.y = r.z;
  r.z = r.w;
  r.w = (r.w ^ (r.w >> 19)) ^ (t ^ (t >> 8));
  return (float)(r.w >> 8) * (1.0f / 16777216.0f);
}

// IntersectRayAABB (bvh_accel.cc:550-593) in its min / max form; q0 = (bmin.x, bmin.y, bmin.z, bmax.x), q1.xy = (bmax.y, bmax.z)
__device__ __forceinline__ bool slab_f(float4 q0, float4 q1, F3 org, F3 inv, float bt) {
  const float ax = (q0.x - org.x) * inv.x, bx = (q0.w - org.x) * inv.x;
  const float ay = (q0.y - org.y) * inv.y, by = (q1.x - org.y) * inv.y;
  const float az = (q0.z - org.z) * inv.z, bz = (q1.y - org.z) * inv.z;
  const float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
  const float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
  return tmax > 0.0f && tmin <= tmax && tmin <= bt;
}

// TriangleIsect (bvh_accel.cc:595-638) on (a0 = p0.xyz e1.x, a1 = e1.yz e2.xy, e2z); a NaN t is rejected
__device__ __forceinline__ void tri_f(float4 a0, float4 a1, float e2z, F3 o, F3 d, uint32_t slot, float &bt, float &bu, float &bv,
                                      uint32_t &bslot) {
  const F3 p0 = f3(a0.x, a0.y, a0.z), e1 = f3(a0.w, a1.x, a1.y), e2 = f3(a1.z, a1.w, e2z);
  const F3 p = fcross(d, e2);
  const float det = fdot(e1, p);
  if (fabsf(det) >= 1.1920929e-07f * 1024.0f * 1.0e-6f) { // fast mode: the reference's guard is relative to its double epsilon
    const float inv = __builtin_amdgcn_rcpf(det);
    const F3 s = o - p0;
    const F3 q = fcross(s, e1);
    const float u = fdot(s, p) * inv, v = fdot(q, d) * inv, t = fdot(e2, q) * inv;
    if (u >= 0.0f && u <= 1.0f && v >= 0.0f && u + v <= 1.0f && t >= 0.0f && t <= bt) {
      bt = t;
      bu = u;
      bv = v;
      bslot = slot;
    }
  }
}

// Plane::intersect (prim-plane.cc:8-44), whose core already is float in the reference
__device__ __forceinline__ bool plane_f(const float pl[4], F3 unit_n, F3 org, F3 dir, float &t_io, F3 &normal) {
  const F3 n = f3(pl[0], pl[1], pl[2]);
  const float vn = fdot(dir, n);
  if (fabsf(vn) > 1.1920929e-07f * 1024.0f) {
    const float t = -(fdot(org, n) + pl[3]) / vn;
    if (t > 0.0f && t < t_io) {
      t_io = t;
      normal = unit_n;
      return true;
    }
  }
  return false;
}

// GenerateBasis + SampleDiffuseIS (render.cc:271-339)
__device__ __forceinline__ F3 sample_diffuse_f(F3 n, Rng &rng) {
  const float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
  const bool x_ok = ax < 1.0e+6f;
  const float m0 = x_ok ? ax : 1.0e+6f;
  const bool y_less = ay < m0;
  const float m1 = y_less ? ay : m0;
  const bool z_less = az < m1;
  const bool use_z = z_less || (!y_less && !x_ok);
  const bool use_y = !z_less && y_less;
  F3 t;
  t.x = use_z ? -n.y : (use_y ? -n.z : 0.0f);
  t.y = use_z ? n.x : (use_y ? 0.0f : -n.z);
  t.z = use_z ? 0.0f : (use_y ? n.x : n.y);
  t = fnormalized(t);
  const F3 b = fnormalized(fcross(t, n));
  const float u1 = rng_next_f(rng), u2 = rng_next_f(rng);
  const float cos_theta = __builtin_sqrtf(1.0f - u1), sin_theta = __builtin_sqrtf(u1); // cos(acos(sqrt(1-u1))), sqrt(1 - cos^2)
  const float sin_phi = __builtin_amdgcn_sinf(u2), cos_phi = __builtin_amdgcn_cosf(u2); // the hardware's argument is in turns
  const F3 T = fscale(t, cos_phi * sin_theta), B = fscale(b, sin_phi * sin_theta), N = fscale(n, cos_theta);
  return (T + B) + N;
}

} // namespace

// ---- scene layout ---------------------------------------------------------------------------------------------------------
__global__ void k_layout_f32(const MgpuNode *__restrict__ nodes, size_t nn, const DTri *__restrict__ tris, size_t nf,
                             const double *__restrict__ slot_normal, int has_fv, const double *__restrict__ mat_diffuse, uint32_t nm,
                             FNode *__restrict__ fnodes, FTri *__restrict__ ftris, float *__restrict__ fnormals, float *__restrict__ fdiffuse) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nn) {
    const MgpuNode n = nodes[i];
    FNode f;
    for (int k = 0; k < 3; ++k) {
      // outward rounding, then 4 ulp more: the slab test itself rounds (two subtractions, two products per axis)
      float lo = __double2float_rd(n.bmin[k]), hi = __double2float_ru(n.bmax[k]);
      lo -= fabsf(lo) * 4.8e-7f + 1.0e-30f;
      hi += fabsf(hi) * 4.8e-7f + 1.0e-30f;
      f.bmin[k] = lo;
      f.bmax[k] = hi;
    }
    if (n.flag == 0) { // interior: children, split axis in the top bits
      f.a = n.data[0];
      f.b = (n.data[1] & 0x3FFFFFFFu) | ((uint32_t)n.axis << 30);
    } else { // leaf: first slot, count
      f.a = n.data[1];
      f.b = (n.data[0] & 0x3FFFFFFFu) | (kLeafTag << 30);
    }
    fnodes[i] = f;
  }
  if (i < nf) {
    const DTri t = tris[i];
    FTri f;
    for (int k = 0; k < 3; ++k) {
      f.v[k] = (float)t.p0[k];
      f.v[3 + k] = (float)t.e1[k];
      f.v[6 + k] = (float)t.e2[k];
    }
    f.mat = t.mat;
    f.pad[0] = f.pad[1] = 0u;
    ftris[i] = f;
    const int per = has_fv ? 9 : 3;
    for (int k = 0; k < per; ++k) fnormals[i * per + k] = (float)slot_normal[i * per + k];
  }
  if (i < 3 * (size_t)nm) fdiffuse[i] = (float)mat_diffuse[i];
}

void launch_layout_f32(hipStream_t s, const MgpuNode *nodes, size_t nn, const DTri *tris, size_t nf, const double *slot_normal,
                       int has_fv, const double *mat_diffuse, uint32_t nm, FNode *fnodes, FTri *ftris, float *fnormals, float *fdiffuse) {
  size_t n = nn > nf ? nn : nf;
  if (n < 3 * (size_t)nm) n = 3 * (size_t)nm;
  if (n == 0) return;
  hipLaunchKernelGGL(k_layout_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nodes, nn, tris, nf, slot_normal, has_fv,
                     mat_diffuse, nm, fnodes, ftris, fnormals, fdiffuse);
}

// ---- the kernel -----------------------------------------------------------------------------------------------------------
template <int CAP, bool LDS_SCENE, int BLOCK, bool OVF>
__global__ __launch_bounds__(BLOCK, (BLOCK == 1024 ? 4 : MGPU_F32_HBM_WAVES(CAP))) void k_render_f32(FScene sc, RenderParams P_arg) {
  __shared__ RenderParams s_P; // launch parameters in LDS, not in scalar registers (mgpu_render_sm.hip)
  if (threadIdx.x == 0) s_P = P_arg;
  __syncthreads();
  const RenderParams &P = s_P;
  MGPU_DYN_SHARED(unsigned char, smem);
  constexpr int kWaves = BLOCK / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  Stack<CAP, OVF> stk;
  stk.lds = reinterpret_cast<uint32_t *>(smem) + ((size_t)wave * CAP) * 64 + lane;
  stk.overflow = (OVF && sc.stack_overflow) ? sc.stack_overflow + gid * sc.overflow_cap : nullptr;

  const unsigned char *lds_nodes = smem + (size_t)kWaves * CAP * 64 * sizeof(uint32_t);
  const unsigned char *lds_tris = lds_nodes + (size_t)P.lds_nodes_bytes;
  if (LDS_SCENE) {
    const uint4 *src = reinterpret_cast<const uint4 *>(sc.nodes);
    uint4 *dst = reinterpret_cast<uint4 *>(const_cast<unsigned char *>(lds_nodes));
    for (uint32_t i = threadIdx.x, n16 = P.lds_nodes_bytes >> 4; i < n16; i += BLOCK) dst[i] = src[i];
    const uint4 *src2 = reinterpret_cast<const uint4 *>(sc.tris);
    uint4 *dst2 = reinterpret_cast<uint4 *>(const_cast<unsigned char *>(lds_tris));
    for (uint32_t i = threadIdx.x, t16 = P.lds_tris_bytes >> 4; i < t16; i += BLOCK) dst2[i] = src2[i];
    __syncthreads();
  }

  const int win_w = P.x1 - P.x0;
  const uint32_t tiles_x = (uint32_t)(win_w + 7) >> 3;
  const uint32_t tiles_y = (uint32_t)(P.n_rows + 7) >> 3;
  const uint32_t total_tiles = tiles_x * tiles_y;
  const uint32_t total_items = total_tiles * (uint32_t)P.passes;
  uint32_t in_item = 64;
  bool exhausted = false;
  constexpr uint32_t kWgChunk = LDS_SCENE ? 16u : 8u;
  const uint32_t shard_items = (total_items + (uint32_t)kShards - 1) / (uint32_t)kShards;
  uint32_t home_shard = 0;
  uint32_t item_tile = 0, item_pass = 0;
  MGPU_XCC_ID(home_shard);
  home_shard &= 7u;
  __shared__ unsigned char s_owner[BLOCK];
  __shared__ unsigned long long wg_cursor;
  __shared__ uint32_t wg_lock, wg_shard_off, wg_dry;
  if (threadIdx.x == 0) {
    wg_cursor = 0ull;
    wg_lock = 0u;
    wg_shard_off = 0u;
    wg_dry = 0u;
  }
  __syncthreads();

  // camera frame and plane in float, once
  const F3 cam_o = f3((float)P.frame[0], (float)P.frame[1], (float)P.frame[2]);
  const F3 plane_n = f3((float)P.plane_n[0], (float)P.plane_n[1], (float)P.plane_n[2]);

  int st = F_SHADE;
  bool have_ray = false, have_path = false;
  uint32_t lx = 0, ly = 0;
  int pass = 0;
  Rng rng{1, 0, 0, 0};
  F3 org = f3(0, 0, 0), dir = f3(0, 0, 1), inv = f3(0, 0, 0);
  float thr0 = 1, thr1 = 1, thr2 = 1;
  int pathLength = 1;
  uint32_t last_mat = kNoMaterial;
  uint32_t cost_base = 0;
  uint32_t sgn = 0; // bit k: dir[k] < 0
  int sp = -1;
  float bt = 3.0e38f, bu = 0, bv = 0;
  uint32_t bslot = kNoHit;
  uint32_t tri_cur = 0, tri_end = 0;
  uint32_t n_rays = 0, n_nodes = 0, n_tris = 0, trace_calls = 0, paths = 0;

  for (;;) {
    const unsigned long long mN = MGPU_BALLOT(st == F_NODE);
    const unsigned long long mT = MGPU_BALLOT(st == F_TRI);
    const unsigned long long mS = MGPU_BALLOT(st == F_SHADE);
    const int cN = __popcll(mN), cT = __popcll(mT), cS = __popcll(mS);
    if ((cN | cT | cS) == 0) break;
    const int cReal = __popcll(MGPU_BALLOT(st == F_SHADE && have_ray));
    const bool run_shade = (cReal >= kShadeMin) || (cN == 0 && cT == 0) || (cS - cReal >= (LDS_SCENE ? 16 : 12));
    if (!run_shade && cN >= 4 * cT) {
      // ================================ NODE step ================================
      if (st == F_NODE) {
#pragma unroll 1
        for (int rep = 0; rep < kNodesPerStep; ++rep) {
          const uint32_t ni = stk.get(sp);
          --sp;
          ++n_nodes;
          float4 q0, q1;
          if (LDS_SCENE) {
            const unsigned char *nd = lds_nodes + (size_t)ni * 32;
            q0 = *reinterpret_cast<const float4 *>(nd);
            q1 = *reinterpret_cast<const float4 *>(nd + 16);
          } else {
            const float4 *nd = reinterpret_cast<const float4 *>(sc.nodes + ni);
            q0 = nd[0];
            q1 = nd[1];
          }
          // the node's last 8 bytes travel with its box (the compiler would fetch them again behind the test)
          MGPU_KEEP2(q1.z, q1.w);
          if (slab_f(q0, q1, org, inv, bt)) {
            const uint32_t a = __float_as_uint(q1.z), b = __float_as_uint(q1.w);
            const uint32_t tag = b >> 30, low = b & 0x3FFFFFFFu;
            if (tag != kLeafTag) {
              const bool nearIsSecond = ((sgn >> tag) & 1u) != 0u; // dirSign[node.axis], bvh_accel.cc:818-824
              stk.put(sp + 1, nearIsSecond ? a : low);   // far
              stk.put(sp + 2, nearIsSecond ? low : a);   // near: popped first
              sp += 2;
            } else if (low != 0u) {
              tri_cur = a;
              tri_end = a + low;
              st = F_TRI;
            }
          }
          if (st != F_NODE || sp < 0) break;
        }
        if (st == F_NODE && sp < 0) st = F_SHADE;
      }
    } else if (!run_shade) {
      // ================================ TRI step =================================
      if (cT <= 32) { // 2 or 4 lanes per open leaf (shared_leaves_step of mgpu_device.hpp, in float; a NaN t is simply rejected)
        const int sh = cT <= 16 ? 2 : 1, m = 1 << sh;
        unsigned char *tbl = s_owner + wave * 64;
        const bool owner = st == F_TRI;
        const uint32_t rank = (uint32_t)__popcll(mT & ((1ull << lane) - 1ull));
        if (owner) tbl[rank] = (unsigned char)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int grp = lane >> sh, sub = lane & (m - 1);
        const bool serving = grp < cT;
        const int own = serving ? (int)tbl[grp] : lane;
        const F3 o = f3(__shfl(org.x, own), __shfl(org.y, own), __shfl(org.z, own));
        const F3 d = f3(__shfl(dir.x, own), __shfl(dir.y, own), __shfl(dir.z, own));
        const uint32_t first = (uint32_t)__shfl((int)tri_cur, own), last = (uint32_t)__shfl((int)tri_end, own);
        float lt = __builtin_inff(), lu = 0.0f, lv = 0.0f;
        uint32_t ls = kNoHit;
        if (serving) {
          uint32_t i = first + (uint32_t)sub;
#pragma unroll 1
          for (int rep = 0; rep < kTrisPerStep && i < last; ++rep, i += (uint32_t)m) {
            float4 a0, a1;
            float e2z;
            if (LDS_SCENE) {
              const unsigned char *tp = lds_tris + (size_t)i * 48;
              a0 = *reinterpret_cast<const float4 *>(tp);
              a1 = *reinterpret_cast<const float4 *>(tp + 16);
              e2z = *reinterpret_cast<const float *>(tp + 32);
            } else {
              const float4 *tp = reinterpret_cast<const float4 *>(sc.tris + i);
              a0 = tp[0];
              a1 = tp[1];
              e2z = sc.tris[i].v[8];
            }
            ++n_tris;
            tri_f(a0, a1, e2z, o, d, i, lt, lu, lv, ls);
          }
        }
        for (int x = 1; x < m; x <<= 1) { // smallest t wins, the later triangle on a tie (what the in-order loop does)
          const float pt = __shfl_xor(lt, x), pu = __shfl_xor(lu, x), pv = __shfl_xor(lv, x);
          const uint32_t ps = (uint32_t)__shfl_xor((int)ls, x);
          const bool take = ps != kNoHit && (ls == kNoHit || pt < lt || (pt == lt && ps > ls));
          if (take) {
            lt = pt;
            lu = pu;
            lv = pv;
            ls = ps;
          }
        }
        const int from = (int)(rank << sh);
        const float ct = __shfl(lt, from), cu = __shfl(lu, from), cv = __shfl(lv, from);
        const uint32_t cs = (uint32_t)__shfl((int)ls, from);
        if (owner) {
          if (cs != kNoHit && ct <= bt) {
            bt = ct;
            bu = cu;
            bv = cv;
            bslot = cs;
          }
          tri_cur += min(tri_end - tri_cur, (uint32_t)(kTrisPerStep << sh));
        }
      } else if (st == F_TRI) {
#pragma unroll 1
        for (int rep = 0; rep < kTrisPerStep; ++rep) {
          float4 a0, a1;
          float e2z;
          if (LDS_SCENE) {
            const unsigned char *tp = lds_tris + (size_t)tri_cur * 48;
            a0 = *reinterpret_cast<const float4 *>(tp);
            a1 = *reinterpret_cast<const float4 *>(tp + 16);
            e2z = *reinterpret_cast<const float *>(tp + 32);
          } else {
            const float4 *tp = reinterpret_cast<const float4 *>(sc.tris + tri_cur);
            a0 = tp[0];
            a1 = tp[1];
            e2z = sc.tris[tri_cur].v[8];
          }
          ++n_tris;
          tri_f(a0, a1, e2z, org, dir, tri_cur, bt, bu, bv, bslot);
          ++tri_cur;
          if (tri_cur == tri_end) break;
        }
      }
      if (st == F_TRI && tri_cur == tri_end) st = sp < 0 ? F_SHADE : F_NODE;
    } else {
      // ================================ SHADE step ===============================
      const bool shade_lane = (st == F_SHADE);
      bool path_done = false, want_pixel = false;
      if (shade_lane) {
        path_done = !have_ray;
        if (have_ray) {
          // ---- the rest of one PathTrace loop iteration (render.cc:403-452) ----
          bool hit = bslot != kNoHit;
          float t = bt;
          F3 n = f3(0, 0, 0);
          if (hit) {
            last_mat = LDS_SCENE ? *reinterpret_cast<const uint32_t *>(lds_tris + (size_t)bslot * 48 + 36) : sc.tris[bslot].mat;
            if (sc.has_fv_normals) { // barycentric lerp, not renormalised (bvh_accel.cc:745-748)
              const float *nn = sc.normals + 9 * (size_t)bslot;
              const float w = 1.0f - bu - bv;
              n.x = w * nn[0] + bu * nn[3] + bv * nn[6];
              n.y = w * nn[1] + bu * nn[4] + bv * nn[7];
              n.z = w * nn[2] + bu * nn[5] + bv * nn[8];
            } else {
              const float *gn = sc.normals + 3 * (size_t)bslot;
              n = f3(gn[0], gn[1], gn[2]);
            }
          }
          if (P.has_plane && plane_f(P.plane, plane_n, org, dir, t, n)) {
            hit = true;
            last_mat = kNoMaterial; // prim-plane.cc:34
          }
          float rad0 = 0.0f, rad1 = 0.0f, rad2 = 0.0f;
          if (!hit) {
            path_done = true;
            if (pathLength < 2) {
              trace_calls += 1;
            } else {
              // the reference's post-miss continuation, in closed loop: same adds, same multiplies, same order
              trace_calls += (uint32_t)P.maxPathLength;
              float d0 = 0.5f, d1 = 0.5f, d2 = 0.5f;
              const bool mul = last_mat != kNoMaterial;
              if (mul && (size_t)(int)last_mat < (size_t)sc.nm) {
                d0 = sc.diffuse[3 * (size_t)last_mat + 0];
                d1 = sc.diffuse[3 * (size_t)last_mat + 1];
                d2 = sc.diffuse[3 * (size_t)last_mat + 2];
              }
              for (int L = pathLength;; ++L) {
                const float il = __builtin_amdgcn_rcpf((float)L) * 0.5f;
                rad0 += thr0 * il;
                rad1 += thr1 * il;
                rad2 += thr2 * il;
                if (L >= P.maxPathLength) break;
                if (mul) { thr0 *= d0; thr1 *= d1; thr2 *= d2; }
              }
            }
          } else if (pathLength >= P.maxPathLength) {
            path_done = true;
            trace_calls += (uint32_t)P.maxPathLength;
          } else {
            const F3 hitP = org + fscale(dir, t);
            (void)rng_next_f(rng); // `double r = randomreal();` drawn and never used (render.cc:430)
            if (fdot(n, fneg(dir)) < 0.0f) n = fneg(n);
            const F3 sd = sample_diffuse_f(n, rng);
            if (last_mat != kNoMaterial) { // Scene::GetMaterial, scene.h:58-65
              if ((size_t)(int)last_mat < (size_t)sc.nm) {
                thr0 *= sc.diffuse[3 * (size_t)last_mat + 0];
                thr1 *= sc.diffuse[3 * (size_t)last_mat + 1];
                thr2 *= sc.diffuse[3 * (size_t)last_mat + 2];
              } else {
                thr0 *= 0.5f; thr1 *= 0.5f; thr2 *= 0.5f;
              }
            }
            org = hitP + fscale(sd, 1.0e-3f);
            dir = sd;
            ++pathLength;
          }
          if (path_done) {
            float *dst = P.pass_stride ? P.out + (size_t)pass * P.pass_stride + ((size_t)(ly >> 3) * tiles_x + (lx >> 3)) * 192u +
                                             (size_t)(((ly & 7u) << 3) + (lx & 7u)) * 3u
                                       : P.out + 3 * ((size_t)ly * (size_t)win_w + lx);
            dst[0] = rad0;
            dst[1] = rad1;
            dst[2] = rad2;
            if (P.tile_cost && pass == 0)
              atomicAdd(P.tile_cost + ((ly >> 3) * tiles_x + (lx >> 3)), n_nodes + n_tris + 16u * n_rays - cost_base);
          }
        }
        have_ray = false;
        want_pixel = path_done;
      }

      // ---- path hand-out, executed by the whole wave (the cursor variables are wave-uniform); see mgpu_render_sm.hip ----
      const bool defer = !exhausted && (cN + cT) > 0 && __popcll(MGPU_BALLOT(want_pixel)) < (LDS_SCENE ? 12 : 8);
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
          const uint32_t slot = in_item + rank;
          if (slot < 64) {
            const uint32_t tx = item_tile % tiles_x, ty = item_tile / tiles_x;
            const uint32_t x = tx * 8 + (slot & 7), y = ty * 8 + (slot >> 3);
            if (x < (uint32_t)win_w && y < (uint32_t)P.n_rows) {
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

      // ---- next path / next traversal ----
      if (shade_lane) {
        if (path_done && have_path) {
          have_path = false;
          // PathTrace prologue (render.cc:387-400)
          const int gy = (P.y_first + (int)(ly / (uint32_t)P.strip_h) * P.y_period + (int)(ly % (uint32_t)P.strip_h)) * P.pix_step;
          const int gx = (P.x0 + (int)lx) * P.pix_step;
          const uint32_t gpix = (uint32_t)gy * (uint32_t)P.W + (uint32_t)gx;
          uint32_t s4[4];
          if (P.rng_mode == MGPU_RNG_TABLE) {
            const uint4 q = reinterpret_cast<const uint4 *>(P.rng_states)[(size_t)pass * P.W * P.H + gpix];
            s4[0] = q.x; s4[1] = q.y; s4[2] = q.z; s4[3] = q.w;
          } else {
            hash_state(P.seed, P.pass_base + (uint32_t)pass, gpix, s4);
          }
          rng = Rng{s4[0], s4[1], s4[2], s4[3]};
          const float u = (float)gx + (rng_next_f(rng) - 0.5f);
          const float v = (float)gy + (rng_next_f(rng) - 0.5f);
          org = cam_o;
          // Camera::GenerateRay (camera.cc:222-240)
          dir = fnormalized(f3((float)(P.frame[3] - P.frame[0]) + u * (float)P.frame[6] + v * (float)P.frame[9],
                               (float)(P.frame[4] - P.frame[1]) + u * (float)P.frame[7] + v * (float)P.frame[10],
                               (float)(P.frame[5] - P.frame[2]) + u * (float)P.frame[8] + v * (float)P.frame[11]));
          thr0 = thr1 = thr2 = 1.0f;
          pathLength = 1;
          ++paths;
          cost_base = n_nodes + n_tris + 16u * n_rays;
          path_done = false;
        }
        if (path_done) {
          if (exhausted) st = F_IDLE;
        } else {
          // BVHAccel::Traverse prologue (bvh_accel.cc:774-802)
          sgn = (dir.x < 0.0f ? 1u : 0u) | (dir.y < 0.0f ? 2u : 0u) | (dir.z < 0.0f ? 4u : 0u);
          inv = f3(__builtin_amdgcn_rcpf(dir.x), __builtin_amdgcn_rcpf(dir.y), __builtin_amdgcn_rcpf(dir.z));
          bt = 3.0e38f; bu = 0.0f; bv = 0.0f; bslot = kNoHit;
          sp = 0;
          stk.put(0, 0u);
          have_ray = true;
          ++n_rays;
          st = F_NODE;
        }
      }
    }
  }

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

template <int CAP, bool LDS, int BLOCK, bool OVF>
static hipError_t launch_f32_one(dim3 grid, hipStream_t s, size_t shmem, const FScene &sc, const RenderParams &p) {
  auto kern = k_render_f32<CAP, LDS, BLOCK, OVF>;
  // one attribute call per device and size (guarded: scenes on different host threads share this table)
  static std::mutex mu;
  static size_t granted[16] = {0};
  if (shmem > 48 * 1024) {
    std::lock_guard<std::mutex> lock(mu);
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || shmem > granted[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) granted[dev] = shmem;
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(BLOCK), shmem, s, sc, p);
  return hipGetLastError();
}

hipError_t launch_render_f32(int cap, bool lds_scene, dim3 grid, hipStream_t s, size_t shmem, const FScene &sc, const RenderParams &p) {
  if (lds_scene) {
    if (cap == 16) return launch_f32_one<16, true, 1024, false>(grid, s, shmem, sc, p);
    if (cap == 24) return launch_f32_one<24, true, 1024, false>(grid, s, shmem, sc, p);
    return hipErrorInvalidValue;
  }
  if (cap == 16) return launch_f32_one<16, false, 256, true>(grid, s, shmem, sc, p);
  if (cap == 24) return launch_f32_one<24, false, 256, true>(grid, s, shmem, sc, p);
  return launch_f32_one<32, false, 256, true>(grid, s, shmem, sc, p);
}

} // namespace mgpu
