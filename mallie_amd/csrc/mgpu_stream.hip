// mgpu_stream.hip -- MGPU_RNG_STREAM: the start state of every eye path in the reference's OWN random stream.
//
// The reference draws all its random numbers from one xorshift128 state per OpenMP thread, consumed in scanline order
// (render.cc:116-168, 657-681); with OMP_NUM_THREADS=1 -- the only deterministic way to run it -- pixel k of a pass
// starts where pixel k-1 stopped.  PathTrace draws 2 numbers for the pixel jitter and then, IF the primary ray hits the
// mesh or the plane, 3 per further iteration up to kMaxPathLength whatever happens on the way: a pixel
// consumes 2 or 2 + 3 (maxPathLength - 1) draws.  So the start state of pixel k is T^(n_k) s0 with
// n_k = 2 k + 3 (maxPathLength - 1) * #{j < k : primary ray of pixel j hits} -- a serial chain, because whether pixel j
// hits depends on its jitter, i.e. on its own start state.
//
// k_stream_states resolves that chain by speculation, one workgroup walking the pixel sequence (all passes, scanline
// order) in windows of 256:
//   1. every lane has a GUESS of its pixel's hit flag; a prefix sum of the guessed draw counts gives each lane its offset
//      from the window's base state; one lane in 16 jumps there with GF(2) matrices T^(2^j) held in LDS (xorshift128 is
//      linear), the others step the generator from their leader's state;
//   2. every lane draws its jitter from that state and traces its primary ray: BVHAccel::Traverse + Plane::intersect,
//      the very functions the render kernels use (mgpu_device.hpp);
//   3. the first lane whose traced flag differs from its guess ends the valid prefix: lanes up to and including it had
//      the right start state (every guess before them was right), so their states are FINAL and go to the table; the
//      base moves behind that lane, the traced flags become the next window's guesses.
// A guess fails only on silhouette pixels, so a window advances by a hundred pixels or more.  The table then feeds the
// ordinary render kernel in MGPU_RNG_TABLE mode; the stream state after the last pixel is handed back to the caller, and
// the next Render() call continues from it as the reference's static state does.
#include <algorithm>
#include <cstring>

#include "mgpu_device.hpp"
#include "mgpu_kernels.hpp"

namespace mgpu {

namespace {
constexpr int kWin = 256;   // pixels per window = threads of the one workgroup
constexpr int kGroup = 16;  // lanes per jump leader
constexpr int kJumpBits = kStreamSerialJumpBits; // offsets inside a window stay below 2^kJumpBits: 256 * (2 + 3 * (maxPathLength - 1))

__device__ __forceinline__ void rng_step(uint32_t s[4]) { // randomreal()'s state update, render.cc:137-168
  const uint32_t t = s[0] ^ (s[0] << 11);
  s[0] = s[1];
  s[1] = s[2];
  s[2] = s[3];
  s[3] = (s[3] ^ (s[3] >> 19)) ^ (t ^ (t >> 8));
}
} // namespace

template <int CAP>
__global__ __launch_bounds__(kWin) void k_stream_states(DScene sc, StreamParams P) {
  __shared__ __attribute__((aligned(16))) uint32_t s_stack[kWin / 64][CAP][64];
  __shared__ uint4 s_jump[kJumpBits][128]; // column i of T^(2^j): the image of unit vector e_i
  __shared__ uint32_t s_scan[kWin];
  __shared__ uint4 s_leader[kWin / kGroup];
  __shared__ unsigned char s_guess[kWin], s_flag[kWin];
  __shared__ uint4 s_base;
  __shared__ int s_first_bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Stack<CAP, true> stk;
  stk.lds = &s_stack[wave][0][lane];
  stk.overflow = sc.stack_overflow ? sc.stack_overflow + (size_t)tid * sc.overflow_cap : nullptr;
  for (int i = tid; i < kJumpBits * 128; i += kWin) (&s_jump[0][0])[i] = P.jump[i];
  if (tid == 0) s_base = *reinterpret_cast<const uint4 *>(P.state);
  s_guess[tid] = 1;
  __syncthreads();
  const unsigned long long npix = (unsigned long long)P.W * (unsigned long long)P.H;
  const unsigned long long total = npix * (unsigned long long)P.passes;
  const uint32_t extra = 3u * (uint32_t)(P.maxPathLength - 1); // draws of a path whose primary ray hits, beyond the two of the jitter
  Counters c{};
  for (unsigned long long frontier = 0; frontier < total;) {
    // 1. offsets from the guesses
    const uint32_t mine = 2u + (s_guess[tid] ? extra : 0u);
    s_scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < kWin; off <<= 1) { // Hillis-Steele inclusive scan, 8 rounds
      const uint32_t v = tid >= off ? s_scan[tid - off] : 0u;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    const uint32_t delta = s_scan[tid] - mine; // exclusive: draws consumed by the window's pixels before mine
    // 2. start state: leaders jump, the others step from their leader
    if ((tid & (kGroup - 1)) == 0) {
      uint4 v = s_base;
      for (int j = 0; j < kJumpBits; ++j)
        if ((delta >> j) & 1u) {
          uint4 acc = make_uint4(0, 0, 0, 0);
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
          for (int i = 0; i < 128; ++i)
            if ((w[i >> 5] >> (i & 31)) & 1u) {
              const uint4 col = s_jump[j][i];
              acc.x ^= col.x; acc.y ^= col.y; acc.z ^= col.z; acc.w ^= col.w;
            }
          v = acc;
        }
      s_leader[tid / kGroup] = v;
    }
    __syncthreads();
    uint32_t st[4];
    {
      const uint4 v = s_leader[tid / kGroup];
      st[0] = v.x; st[1] = v.y; st[2] = v.z; st[3] = v.w;
      const uint32_t lead_delta = s_scan[tid & ~(kGroup - 1)] - (2u + (s_guess[tid & ~(kGroup - 1)] ? extra : 0u));
      for (uint32_t k = delta - lead_delta; k; --k) rng_step(st);
    }
    // 3. the primary ray of my pixel under that state
    const unsigned long long q = frontier + (unsigned long long)tid;
    const bool live = q < total;
    bool hit = false;
    if (live) {
      const uint32_t pix = (uint32_t)(q % npix);
      const int gx = (int)(pix % (uint32_t)P.W), gy = (int)(pix / (uint32_t)P.W);
      Rng rng{st[0], st[1], st[2], st[3]};
      const float ju = (float)(rng_next(rng) - 0.5);
      const float jv = (float)(rng_next(rng) - 0.5);
      const V3 org = v3(P.frame[0], P.frame[1], P.frame[2]);
      const V3 dir = camera_dir(P.frame, (double)((float)gx + ju), (double)((float)gy + jv));
      Hit h;
      traverse<CAP, true>(sc, stk, org, dir, h, c);
      hit = h.t < kDblMax; // bvh_accel.cc:838
      if (P.has_plane) {
        double t = h.t;
        V3 n;
        if (plane_hit(P.plane, P.plane_n, org, dir, t, n)) hit = true;
      }
    }
    s_flag[tid] = hit ? 1 : 0;
    if (tid == 0) s_first_bad = kWin;
    __syncthreads();
    if (live && (hit ? 1 : 0) != s_guess[tid]) atomicMin(&s_first_bad, tid);
    __syncthreads();
    // 4. the valid prefix: everything up to and including the first wrong guess
    const unsigned long long left = total - frontier;
    int valid = s_first_bad + 1;
    if (valid > kWin) valid = kWin;
    if ((unsigned long long)valid > left) valid = (int)left;
    if (tid < valid) {
      reinterpret_cast<uint4 *>(P.table)[q] = make_uint4(st[0], st[1], st[2], st[3]);
      if (tid == valid - 1) { // the base moves behind me: my start state advanced by what my path really draws
        for (uint32_t k = 2u + (hit ? extra : 0u); k; --k) rng_step(st);
        s_base = make_uint4(st[0], st[1], st[2], st[3]);
      }
    }
    __syncthreads();
    // 5. next window's guesses: the flags just traced for the pixels that stay, their neighbour's for the new ones
    const int from = tid + valid;
    const unsigned char g = from < kWin ? s_flag[from] : s_flag[kWin - 1];
    __syncthreads();
    s_guess[tid] = g;
    frontier += (unsigned long long)valid;
    __syncthreads();
  }
  if (tid == 0) *reinterpret_cast<uint4 *>(P.state) = s_base;
}

hipError_t launch_stream_states(int cap, hipStream_t s, const DScene &sc, const StreamParams &p) {
  switch (cap) {
  case 16: hipLaunchKernelGGL(k_stream_states<16>, dim3(1), dim3(kWin), 0, s, sc, p); break;
  case 24: hipLaunchKernelGGL(k_stream_states<24>, dim3(1), dim3(kWin), 0, s, sc, p); break;
  default: hipLaunchKernelGGL(k_stream_states<32>, dim3(1), dim3(kWin), 0, s, sc, p); break;
  }
  return hipGetLastError();
}

// =====================================================================================================================
// Chip-wide resolution (round 4).  k_stream_states above walks the chain with ONE workgroup -- 0.97 s per 1080p pass on 1 / 256
// of the chip.  What makes the chain serial is only this: the start state of pixel k is T^(2k + E h_k) s0, E = 3 (maxPathLength
// - 1), h_k = primary rays that hit among pixels 0 .. k-1, and whether pixel k hits depends on its jitter, i.e. on h_k.  But
// for all pixels except those on the silhouette of (mesh + plane) against the sky the flag is the same for EVERY jitter, and for
// a silhouette pixel it is a function of h_k alone.  So:
//
//   classify (once per camera)  every pixel's primary ray at the four corners and the centre of its jitter square: five equal
//                               flags -> "certain" (hit / miss), else "uncertain".  A guess, verified below -- never trusted.
//   scan                        C[k] = certain hits before pixel k, J[k] = uncertain pixels before k, U[j] = the j-th uncertain pixel
//   per pass:
//     bases                     b_j = T^(2 U[j] + E C[U[j]]) s0 for every uncertain pixel, all at once (what its start state is short
//                               of the uncertain hits before it)
//     rounds of L = 128         the chain over the uncertain pixels only.  Inside a round the number S of uncertain hits so far
//                               can only be S0 .. S0 + i at the round's i-th pixel: ALL L (L + 1) / 2 candidates (i, s) are traced
//                               at once -- state T^(E (S0 + s)) b_j, jitter, primary ray -- into a table F; then one lane walks
//                               s += F[i][s] through it: L table reads instead of L ray casts on the critical path.  The walk of
//                               round r is done by every workgroup of round r + 1's launch (which needs its S0).
//     finish                    h_k = C[k] + (uncertain hits before k) for EVERY pixel, its start state into the table -- a thread jumps
//                               from s0 to the first pixel of its run and steps the generator from pixel to pixel (round 6) -- and the
//                               VERIFICATION: every pixel's primary ray is traced with its final jitter and
//                               compared with the flag the chain assumed.  A "certain" pixel that disagrees (a feature smaller than
//                               a pixel that the five probes missed) becomes uncertain and everything is resolved again; pixels
//                               stay uncertain for later calls (the classification is cached per camera with the scene).
//   The table equals the serial kernel's word for word (tests), 1080p pass: see DESIGN.md 4.7.
// =====================================================================================================================
#ifndef MGPU_STREAM_FINISH_RUN
#define MGPU_STREAM_FINISH_RUN 4
#endif
namespace {
constexpr int kRoundL = 128;                           // uncertain pixels per round
constexpr int kRoundCand = kRoundL * (kRoundL + 1) / 2; // candidates (i, s), 0 <= s <= i < L
constexpr int kSBlock = 256;                           // threads per workgroup of the kernels below
constexpr int kClassifyRandom = 11;                    // random probes per pixel beside the five fixed ones
constexpr uint32_t kFinishRun = MGPU_STREAM_FINISH_RUN; // fewest consecutive pixels a thread of k_stream_finish walks from one jump
constexpr int kPromoteDx = 8, kPromoteDy = 1;          // a pixel that fails its verification takes this neighbourhood with it

// v <- T^n v with the columns of T^(2^j) in LDS (`jm`: [levels][128])
__device__ __forceinline__ uint4 stream_jump(uint4 v, unsigned long long n, const uint4 *jm) {
  for (int j = 0; n; ++j, n >>= 1) {
    if (!(n & 1ull)) continue;
    const uint4 *col = jm + (size_t)j * 128;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t bits = w[k];
      while (bits) {
        const int b = __ffs((int)bits) - 1;
        bits &= bits - 1u;
        const uint4 c = col[32 * k + b];
        acc.x ^= c.x; acc.y ^= c.y; acc.z ^= c.z; acc.w ^= c.w;
      }
    }
    v = acc;
  }
  return v;
}

// Does the primary ray of pixel (gx, gy) through sample position (gx + ju, gy + jv) hit the mesh or the plane?  PathTrace's first
// iteration up to the test that decides how many numbers the path draws (render.cc:387-408); the very functions of the render kernels.
template <int CAP>
__device__ __forceinline__ bool primary_hits(const DScene &sc, const Stack<CAP, true> &stk, const StreamParams &P, int gx, int gy, float ju, float jv,
                                             Counters &c) {
  const V3 org = v3(P.frame[0], P.frame[1], P.frame[2]);
  const V3 dir = camera_dir(P.frame, (double)((float)gx + ju), (double)((float)gy + jv));
  Hit h;
  traverse<CAP, true>(sc, stk, org, dir, h, c);
  bool hit = h.t < kDblMax; // bvh_accel.cc:838
  if (P.has_plane) {
    double t = h.t;
    V3 n;
    if (plane_hit(P.plane, P.plane_n, org, dir, t, n)) hit = true;
  }
  return hit;
}
template <int CAP>
__device__ __forceinline__ bool primary_hits_state(const DScene &sc, const Stack<CAP, true> &stk, const StreamParams &P, uint32_t pix, uint4 st,
                                                   Counters &c) {
  Rng rng{st.x, st.y, st.z, st.w};
  const float ju = (float)(rng_next(rng) - 0.5);
  const float jv = (float)(rng_next(rng) - 0.5);
  return primary_hits<CAP>(sc, stk, P, (int)(pix % (uint32_t)P.W), (int)(pix / (uint32_t)P.W), ju, jv, c);
}

template <int CAP> struct StreamLds {
  uint32_t stack[kSBlock / 64][CAP][64];
};
template <int CAP>
__device__ __forceinline__ Stack<CAP, true> bind_stack(StreamLds<CAP> &l, const DScene &sc) {
  Stack<CAP, true> stk;
  stk.lds = &l.stack[threadIdx.x >> 6][0][threadIdx.x & 63];
  stk.overflow = sc.stack_overflow ? sc.stack_overflow + ((size_t)blockIdx.x * kSBlock + threadIdx.x) * sc.overflow_cap : nullptr;
  return stk;
}
__device__ __forceinline__ void load_jump(uint4 *jm, const uint4 *src, int levels) {
  for (int i = threadIdx.x; i < levels * 128; i += kSBlock) jm[i] = src[i];
  __syncthreads();
}

// ---- classify: 0 = every probe misses, 1 = every probe hits, 2 = uncertain -------------------------------------------------
template <int CAP>
__global__ __launch_bounds__(kSBlock) void k_stream_classify(DScene sc, StreamParams P, unsigned char *__restrict__ cls) {
  __shared__ StreamLds<CAP> lds;
  const Stack<CAP, true> stk = bind_stack(lds, sc);
  const uint32_t npix = (uint32_t)P.W * (uint32_t)P.H;
  Counters c{};
  for (uint32_t pix = blockIdx.x * kSBlock + threadIdx.x; pix < npix; pix += gridDim.x * kSBlock) {
    const int gx = (int)(pix % (uint32_t)P.W), gy = (int)(pix / (uint32_t)P.W);
    int hits = 0;
    const float a[5] = {-0.5f, 0.5f, -0.5f, 0.5f, 0.0f}, b[5] = {-0.5f, -0.5f, 0.5f, 0.5f, 0.0f};
    // The plane's horizon, analytically (round-4 review item 4).  Plane::intersect accepts a ray iff |v.n| > 1024 FLT_EPSILON and
    // t = -(o.n + d) / (v.n) > 0 (prim-plane.cc:8-44): both conditions change only where v.n passes through the band
    // [-1024 eps, +1024 eps].  v.n is a smooth function of the sample position, all but linear over one pixel: when the interval its
    // four corner values span, widened by a sixteenth of its width, reaches that band, some jitter of this pixel may be accepted and another
    // refused -- the pixel is uncertain from the first call on, whatever the sixteen probes below would have said (they found
    // the 0.08-pixel band of the reference's default view one failed verification at a time).  Marking more pixels uncertain is
    // always safe: an uncertain pixel's flag is traced, not assumed.
#ifndef MGPU_STREAM_NO_HORIZON // (A/B switch: the round-4 classification, probes only)
    if (P.has_plane) {
      const V3 n = v3((double)P.plane[0], (double)P.plane[1], (double)P.plane[2]);
      float lo = __builtin_inff(), hi = -__builtin_inff();
      for (int k = 0; k < 4; ++k) {
        const V3 d = camera_dir(P.frame, (double)((float)gx + a[k]), (double)((float)gy + b[k]));
        const float vn = (float)dot(normalized_w(d), n); // plane_hit's own expression
        lo = fminf(lo, vn);
        hi = fmaxf(hi, vn);
      }
      const float thr = 1.1920929e-07f * 1024.0f, pad = 0.0625f * (hi - lo) + 1.0e-9f; // (second-order terms of v.n across a pixel are ~1e-3 of its span; the float rounding of v.n ~1e-11)
      if (!(lo - pad > thr) && !(hi + pad < -thr)) { // (a NaN lands here too)
        cls[pix] = 2;
        continue;
      }
    }
#endif
    for (int k = 0; k < 5; ++k) hits += primary_hits<CAP>(sc, stk, P, gx, gy, a[k], b[k], c) ? 1 : 0;
    // ... and kClassifyRandom jitters of the pixel's own: features thinner than a pixel that touch neither a corner nor the centre --
    // seen at once on the reference's default view: the eye sits at the height of the Cornell box's floor, and between the walls'
    // lower edge and the direction below which Plane::intersect accepts a ray (|v.n| > 1024 FLT_EPSILON) a band 0.08 pixels high
    // of the horizon row misses everything
    Rng pr{pix * 2654435761u + 1u, pix ^ 0x9E3779B9u, 0x85EBCA6Bu + pix * 40503u, 0xC2B2AE35u};
    for (int k = 0; k < 4; ++k) (void)rng_next_u32(pr);
    for (int k = 0; k < kClassifyRandom && (hits == 0 || hits == 5 + k); ++k) {
      const float ju = (float)(rng_next(pr) - 0.5), jv = (float)(rng_next(pr) - 0.5);
      hits += primary_hits<CAP>(sc, stk, P, gx, gy, ju, jv, c) ? 1 : 0;
    }
    cls[pix] = hits == 0 ? 0 : (hits == 5 + kClassifyRandom ? 1 : 2);
  }
}

// ---- exclusive scan of (certain hit, uncertain) counts, 1024 pixels per workgroup --------------------------------------------
// class of a pixel: 0 / 1 = certain miss / hit, 2 = uncertain; 4 + (0 / 1) = a certain pixel whose verification failed in the
// attempt that is running (still counted as certain by that attempt's C / J; uncertain from the next scan on)
__device__ __forceinline__ unsigned long long cls_value(unsigned char c) { return c == 1 ? 1ull : (c >= 2 ? (1ull << 32) : 0ull); }
__device__ __forceinline__ unsigned long long block_scan_excl(unsigned long long v, unsigned long long *s_wave, unsigned long long &total) {
  // exclusive scan over the 256 threads of the workgroup
  unsigned long long incl = v;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned long long base = 0;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  __syncthreads();
  return base + incl - v;
}
__global__ __launch_bounds__(kSBlock) void k_stream_scan_partial(const unsigned char *__restrict__ cls, uint32_t npix, unsigned long long *__restrict__ block_sum) {
  __shared__ unsigned long long s_wave[4];
  const uint32_t p0 = (blockIdx.x * kSBlock + threadIdx.x) * 4u;
  unsigned long long v = 0;
  for (uint32_t k = 0; k < 4; ++k)
    if (p0 + k < npix) v += cls_value(cls[p0 + k]);
  unsigned long long total;
  (void)block_scan_excl(v, s_wave, total);
  if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}
__global__ __launch_bounds__(kSBlock) void k_stream_scan_blocks(unsigned long long *__restrict__ block_sum, uint32_t n_blocks, uint32_t *__restrict__ totals) {
  __shared__ unsigned long long s_wave[4];
  unsigned long long carry = 0;
  for (uint32_t b0 = 0; b0 < n_blocks; b0 += kSBlock) {
    const uint32_t b = b0 + threadIdx.x;
    const unsigned long long v = b < n_blocks ? block_sum[b] : 0ull;
    unsigned long long total;
    const unsigned long long ex = block_scan_excl(v, s_wave, total);
    if (b < n_blocks) block_sum[b] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) {
    totals[0] = (uint32_t)(carry >> 32); // m: uncertain pixels
    totals[1] = (uint32_t)carry;         // certain hits of a pass
  }
}
__global__ __launch_bounds__(kSBlock) void k_stream_scan_final(unsigned char *__restrict__ cls, uint32_t npix, const unsigned long long *__restrict__ block_off,
                                                              uint32_t *__restrict__ C, uint32_t *__restrict__ J, uint32_t *__restrict__ U) {
  __shared__ unsigned long long s_wave[4];
  const uint32_t p0 = (blockIdx.x * kSBlock + threadIdx.x) * 4u;
  unsigned char c[4] = {0, 0, 0, 0};
  unsigned long long v = 0;
  for (uint32_t k = 0; k < 4; ++k)
    if (p0 + k < npix) {
      c[k] = cls[p0 + k];
      v += cls_value(c[k]);
    }
  for (uint32_t k = 0; k < 4; ++k)
    if (p0 + k < npix && c[k] > 2) cls[p0 + k] = 2; // promoted by the last attempt's verification
  unsigned long long total;
  unsigned long long run = block_off[blockIdx.x] + block_scan_excl(v, s_wave, total);
  for (uint32_t k = 0; k < 4; ++k)
    if (p0 + k < npix) {
      C[p0 + k] = (uint32_t)run;
      J[p0 + k] = (uint32_t)(run >> 32);
      if (c[k] >= 2) U[(uint32_t)(run >> 32)] = p0 + k;
      run += cls_value(c[k]);
    }
}

struct RoundParams {
  int levels;                // jump matrices in use
  uint32_t E;                // 3 * (maxPathLength - 1)
  uint32_t m;                // uncertain pixels per pass
  const uint32_t *U, *C;     // the j-th uncertain pixel; certain hits before a pixel
  uint4 *base;               // [m]
  unsigned char *F[2];       // candidate flags of the even / odd rounds, kRoundCand bytes each
  uint32_t *Sarr;            // [rounds + 1]: uncertain hits before round r (Sarr[0] = 0 by memset)
  uint32_t *USx;             // [m + 1]: uncertain hits before the j-th uncertain pixel; [m] = all of them
  unsigned char *uflag;      // [m]
};

// ---- bases ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSBlock) void k_stream_bases(StreamParams P, RoundParams R) {
  MGPU_DYN_SHARED(uint4, jm);
  load_jump(jm, P.jump, R.levels);
  const uint4 s0 = *reinterpret_cast<const uint4 *>(P.state);
  for (uint32_t j = blockIdx.x * kSBlock + threadIdx.x; j < R.m; j += gridDim.x * kSBlock) {
    const uint32_t pix = R.U[j];
    R.base[j] = stream_jump(s0, 2ull * pix + (unsigned long long)R.E * R.C[pix], jm);
  }
}

// ---- one round: walk the previous round's table, then trace this round's candidates ---------------------------------------------
template <int CAP>
__global__ __launch_bounds__(kSBlock) void k_stream_round(DScene sc, StreamParams P, RoundParams R, uint32_t r) {
  MGPU_DYN_SHARED(uint4, jm);
  __shared__ StreamLds<CAP> lds;
  __shared__ unsigned char s_F[kRoundCand];
  __shared__ uint32_t s_before[kRoundL + 1]; // hits before the round's i-th pixel, relative to the round's start
  __shared__ uint32_t s_S0;
  load_jump(jm, P.jump, R.levels);
  uint32_t S0 = 0;
  if (r > 0) {
    const uint32_t j0 = (r - 1) * (uint32_t)kRoundL;
    const uint32_t cnt = min((uint32_t)kRoundL, R.m - j0);
    const unsigned char *Fp = R.F[(r - 1) & 1];
    for (uint32_t i = threadIdx.x; i < cnt * (cnt + 1) / 2; i += kSBlock) s_F[i] = Fp[i];
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t s = 0;
      for (uint32_t i = 0; i < cnt; ++i) {
        s_before[i] = s;
        s += s_F[i * (i + 1) / 2 + s];
      }
      s_before[cnt] = s;
      s_S0 = R.Sarr[r - 1] + s;
    }
    __syncthreads();
    S0 = s_S0;
    if (blockIdx.x == 0) { // one workgroup publishes what the walk found
      const uint32_t prev = R.Sarr[r - 1];
      for (uint32_t i = threadIdx.x; i < cnt; i += kSBlock) {
        R.USx[j0 + i] = prev + s_before[i];
        R.uflag[j0 + i] = (unsigned char)(s_before[i + 1] - s_before[i]);
      }
      if (threadIdx.x == 0) {
        R.Sarr[r] = S0;
        if (j0 + cnt == R.m) R.USx[R.m] = S0;
      }
    }
  } else if (blockIdx.x == 0 && threadIdx.x == 0 && R.m == 0) {
    R.USx[0] = 0;
  }
  const uint32_t j0 = r * (uint32_t)kRoundL;
  if (j0 >= R.m) return; // the launch behind the last round only walks
  const uint32_t cnt = min((uint32_t)kRoundL, R.m - j0);
  const uint32_t ncand = cnt * (cnt + 1) / 2;
  const Stack<CAP, true> stk = bind_stack(lds, sc);
  Counters c{};
  unsigned char *Fo = R.F[r & 1];
  for (uint32_t q = blockIdx.x * kSBlock + threadIdx.x; q < ncand; q += gridDim.x * kSBlock) {
    // q = i (i + 1) / 2 + s
    uint32_t i = (uint32_t)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
    while (i * (i + 1) / 2 > q) --i;
    while ((i + 1) * (i + 2) / 2 <= q) ++i;
    const uint32_t sdelta = q - i * (i + 1) / 2;
    const uint4 st = stream_jump(R.base[j0 + i], (unsigned long long)R.E * (S0 + sdelta), jm);
    Fo[q] = primary_hits_state<CAP>(sc, stk, P, R.U[j0 + i], st, c) ? 1 : 0;
  }
}

// ---- finish a pass: every pixel's start state into the table, and the verification -----------------------------------------------
template <int CAP>
__global__ __launch_bounds__(kSBlock) void k_stream_finish(DScene sc, StreamParams P, RoundParams R, unsigned char *__restrict__ cls, const uint32_t *__restrict__ J,
                                                          uint32_t pass, uint32_t *__restrict__ bad /* [0] certain pixels that disagree, [1] uncertain ones */) {
  MGPU_DYN_SHARED(uint4, jm);
  __shared__ StreamLds<CAP> lds;
  load_jump(jm, P.jump, R.levels);
  const Stack<CAP, true> stk = bind_stack(lds, sc);
  const uint32_t npix = (uint32_t)P.W * (uint32_t)P.H;
  const uint4 s0 = *reinterpret_cast<const uint4 *>(P.state);
  uint4 *table = reinterpret_cast<uint4 *>(P.table) + (size_t)pass * npix;
  Counters c{};
  // A thread takes a RUN of consecutive pixels: it jumps from s0 to the first pixel's state once (a GF(2) matrix product per set bit
  // of the offset: ~7 000 instructions) and walks on from there -- the next pixel starts 2 or 2 + E draws later, i.e. 16 or 16 + 8 E
  // instructions of the generator itself.  (Round 6, counted on the ISA interpreter: the per-pixel jumps were 92 % of this kernel's and 60 % of a
  // settled 1080p resolution's vector instructions.)  The offsets are still taken from C / USx pixel by pixel: a step that is neither of the two
  // (it cannot be) jumps.
  // Run length: the grid's threads share the frame evenly, one run each (a 1080p pass on 256 CUs: 8 pixels), at least kFinishRun -- this
  // kernel is short enough that the longest thread decides its time.
  const uint32_t nthreads = gridDim.x * kSBlock;
  const uint32_t run_len = max(kFinishRun, (npix + nthreads - 1) / nthreads);
  const uint32_t nruns = (npix + run_len - 1) / run_len;
  for (uint32_t run = blockIdx.x * kSBlock + threadIdx.x; run < nruns; run += gridDim.x * kSBlock) {
    const uint32_t p0 = run * run_len, p1 = min(npix, p0 + run_len);
    unsigned long long n_cur = 0;
    uint4 st = make_uint4(0, 0, 0, 0);
    for (uint32_t pix = p0; pix < p1; ++pix) {
      unsigned char k = cls[pix];
      if (k >= 4) k -= 4; // promoted by an earlier pass of this attempt: this attempt's C / J still count it as certain
      const uint32_t j = J[pix];
      const uint32_t h = R.C[pix] + R.USx[j];
      const unsigned long long n = 2ull * pix + (unsigned long long)R.E * h;
      if (pix == p0 || n < n_cur || n - n_cur > 2ull + R.E) st = stream_jump(s0, n, jm);
      else {
        uint32_t w[4] = {st.x, st.y, st.z, st.w};
        for (uint32_t d = (uint32_t)(n - n_cur); d; --d) rng_step(w);
        st = make_uint4(w[0], w[1], w[2], w[3]);
      }
      n_cur = n;
      table[pix] = st;
      const bool assumed = k == 1 || (k == 2 && R.uflag[j] != 0);
      const bool truth = primary_hits_state<CAP>(sc, stk, P, pix, st, c);
      if (truth != assumed) {
        if (k == 2) atomicAdd(&bad[1], 1u); // cannot happen: the chain traced exactly this ray
        else {
          // the probes missed something smaller than a pixel: uncertain from the next attempt on (and for later calls) -- together with
          // its certain neighbours: such features come in runs (an edge seen edge-on), and every attempt costs a whole resolution
          const int gx = (int)(pix % (uint32_t)P.W), gy = (int)(pix / (uint32_t)P.W);
          for (int dy = -kPromoteDy; dy <= kPromoteDy; ++dy)
            for (int dx = -kPromoteDx; dx <= kPromoteDx; ++dx) {
              const int x = gx + dx, y = gy + dy;
              if (x < 0 || y < 0 || x >= P.W || y >= P.H) continue;
              const uint32_t q = (uint32_t)y * (uint32_t)P.W + (uint32_t)x;
              const unsigned char cq = cls[q];
              if (cq < 2) cls[q] = 4 + cq; // (idempotent: whoever else promotes it writes the same value)
            }
          atomicAdd(&bad[0], 1u);
        }
      }
    }
  }
}

// ---- the stream state behind a pass ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_stream_advance(StreamParams P, RoundParams R, const uint32_t *__restrict__ totals) {
  MGPU_DYN_SHARED(uint4, jm);
  for (int i = threadIdx.x; i < R.levels * 128; i += 64) jm[i] = P.jump[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long npix = (unsigned long long)P.W * (unsigned long long)P.H;
    const unsigned long long hits = (unsigned long long)totals[1] + R.USx[R.m];
    uint4 *st = reinterpret_cast<uint4 *>(P.state);
    *st = stream_jump(*st, 2ull * npix + (unsigned long long)R.E * hits, jm);
  }
}

template <typename K> hipError_t grant_lds(K kern, size_t dyn) {
  return dyn > 32 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) : hipSuccess;
}

#define STRY(expr)                    \
  do {                                \
    hipError_t e_ = (expr);           \
    if (e_ != hipSuccess) return e_;  \
  } while (0)

template <int CAP>
hipError_t resolve_cap(hipStream_t st, const DScene &sc, StreamParams P, StreamScratch &X, int num_cu, bool fresh_camera, uint32_t *retries_out,
                       bool *unsettled_out) {
  const uint32_t npix = (uint32_t)P.W * (uint32_t)P.H;
  const uint32_t E = 3u * (uint32_t)(P.maxPathLength - 1);
  // jump levels: the largest exponent is a whole pass's draws
  const unsigned long long max_n = (2ull + E) * npix + 2ull;
  int levels = 1;
  while (levels < kStreamJumpBits && (max_n >> levels)) ++levels;
  if (max_n >> levels) return hipErrorInvalidValue;
  const size_t dyn = (size_t)levels * 128 * sizeof(uint4);
  STRY(grant_lds(k_stream_bases, dyn));
  STRY(grant_lds(k_stream_round<CAP>, dyn));
  STRY(grant_lds(k_stream_finish<CAP>, dyn));
  STRY(grant_lds(k_stream_advance, dyn));
  const uint32_t n_blocks = (npix + 1023u) / 1024u;
  const int wide = num_cu * 4; // workgroups of the grid-stride kernels
  uint4 state0;
  STRY(hipMemcpyAsync(&state0, P.state, 16, hipMemcpyDeviceToHost, st));
  STRY(hipStreamSynchronize(st));
  if (fresh_camera) hipLaunchKernelGGL(k_stream_classify<CAP>, dim3(wide), dim3(kSBlock), 0, st, sc, P, X.cls);
  uint32_t retries = 0;
  for (;; ++retries) {
    if (retries > 64) { // every retry makes at least one more pixel uncertain; this is not convergence trouble
      *unsettled_out = true; // (a verdict of its own, not a HIP error code: the caller falls back to the one-workgroup walk)
      if (retries_out) *retries_out = retries;
      return hipSuccess;
    }
    hipLaunchKernelGGL(k_stream_scan_partial, dim3(n_blocks), dim3(kSBlock), 0, st, X.cls, npix, X.block_sum);
    hipLaunchKernelGGL(k_stream_scan_blocks, dim3(1), dim3(kSBlock), 0, st, X.block_sum, n_blocks, X.totals);
    hipLaunchKernelGGL(k_stream_scan_final, dim3(n_blocks), dim3(kSBlock), 0, st, X.cls, npix, X.block_sum, X.C, X.J, X.U);
    uint32_t totals[2];
    STRY(hipMemcpyAsync(totals, X.totals, 8, hipMemcpyDeviceToHost, st));
    STRY(hipMemsetAsync(X.bad, 0, 8, st));
    STRY(hipStreamSynchronize(st));
    const uint32_t m = totals[0];
    const uint32_t rounds = (m + (uint32_t)kRoundL - 1) / (uint32_t)kRoundL;
    if ((size_t)rounds + 1 > X.sarr_cap) return hipErrorOutOfMemory; // sized for every pixel uncertain: cannot happen
    RoundParams R;
    R.levels = levels; R.E = E; R.m = m; R.U = X.U; R.C = X.C; R.base = X.base; R.F[0] = X.F; R.F[1] = X.F + kRoundCand;
    R.Sarr = X.Sarr; R.USx = X.USx; R.uflag = X.uflag;
    const int round_grid = (kRoundCand + kSBlock - 1) / kSBlock;
    for (int pass = 0; pass < P.passes; ++pass) {
      STRY(hipMemsetAsync(X.Sarr, 0, 4, st));
      if (m) hipLaunchKernelGGL(k_stream_bases, dim3(std::min<uint32_t>((m + kSBlock - 1) / kSBlock, (uint32_t)wide)), dim3(kSBlock), dyn, st, P, R);
      for (uint32_t r = 0; r <= rounds; ++r)
        hipLaunchKernelGGL(k_stream_round<CAP>, dim3(r < rounds ? round_grid : 1), dim3(kSBlock), dyn, st, sc, P, R, r);
      hipLaunchKernelGGL(k_stream_finish<CAP>, dim3(wide), dim3(kSBlock), dyn, st, sc, P, R, X.cls, X.J, (uint32_t)pass, X.bad);
      hipLaunchKernelGGL(k_stream_advance, dim3(1), dim3(64), dyn, st, P, R, X.totals);
    }
    STRY(hipGetLastError());
    uint32_t bad[2];
    STRY(hipMemcpyAsync(bad, X.bad, 8, hipMemcpyDeviceToHost, st));
    STRY(hipStreamSynchronize(st));
    if (bad[1]) return hipErrorAssert; // an uncertain pixel whose resolved flag is not its traced flag: a bug, not a retry
    if (!bad[0]) break;
    STRY(hipMemcpyAsync(P.state, &state0, 16, hipMemcpyHostToDevice, st)); // again, from the stream state the call came with
  }
  if (retries_out) *retries_out = retries;
  return hipSuccess;
}
#undef STRY
} // namespace

size_t stream_scratch_sarr_cap(size_t npix) { return (npix + kRoundL - 1) / kRoundL + 2; }
size_t stream_scratch_f_bytes() { return 2 * (size_t)kRoundCand; }

hipError_t stream_states_resolve(int cap, hipStream_t st, const DScene &sc, const StreamParams &p, StreamScratch &scratch, int num_cu, bool fresh_camera,
                                 uint32_t *retries_out, bool *unsettled_out) {
  *unsettled_out = false;
  switch (cap) {
  case 16: return resolve_cap<16>(st, sc, p, scratch, num_cu, fresh_camera, retries_out, unsettled_out);
  case 24: return resolve_cap<24>(st, sc, p, scratch, num_cu, fresh_camera, retries_out, unsettled_out);
  default: return resolve_cap<32>(st, sc, p, scratch, num_cu, fresh_camera, retries_out, unsettled_out);
  }
}

// Columns of T^(2^j), j = 0..kStreamJumpBits-1, over GF(2): column i of T is the generator's update applied to the state
// with only bit i set (the update is linear: shifts and xors); squaring a matrix = applying it to its own columns.
void stream_jump_matrices(uint32_t *out /* kStreamJumpBits * 128 * 4 words */) {
  auto apply = [](const uint32_t (*m)[4], const uint32_t v[4], uint32_t r[4]) {
    r[0] = r[1] = r[2] = r[3] = 0;
    for (int i = 0; i < 128; ++i)
      if ((v[i >> 5] >> (i & 31)) & 1u)
        for (int k = 0; k < 4; ++k) r[k] ^= m[i][k];
  };
  static uint32_t cur[128][4], nxt[128][4];
  for (int i = 0; i < 128; ++i) {
    uint32_t s[4] = {0, 0, 0, 0};
    s[i >> 5] = 1u << (i & 31);
    const uint32_t t = s[0] ^ (s[0] << 11);
    cur[i][0] = s[1];
    cur[i][1] = s[2];
    cur[i][2] = s[3];
    cur[i][3] = (s[3] ^ (s[3] >> 19)) ^ (t ^ (t >> 8));
  }
  for (int j = 0; j < kStreamJumpBits; ++j) {
    memcpy(out + (size_t)j * 128 * 4, cur, sizeof(cur));
    for (int i = 0; i < 128; ++i) apply(cur, cur[i], nxt[i]);
    memcpy(cur, nxt, sizeof(cur));
  }
}

} // namespace mgpu
