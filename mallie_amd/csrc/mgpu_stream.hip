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
#include <cstring>

#include "mgpu_device.hpp"
#include "mgpu_kernels.hpp"

namespace mgpu {

namespace {
constexpr int kWin = 256;   // pixels per window = threads of the one workgroup
constexpr int kGroup = 16;  // lanes per jump leader
constexpr int kJumpBits = kStreamJumpBits; // offsets inside a window stay below 2^kJumpBits: 256 * (2 + 3 * (maxPathLength - 1))

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
