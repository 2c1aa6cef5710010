// mgpu_render_sm.hip -- k_render_sm: the wave-scheduled ("state machine") persistent path tracer for gfx950.
//
// Same per-path arithmetic as PathTrace / BVHAccel::Traverse (see mgpu_device.hpp for the contract); what changes is
// HOW a 64-lane wave walks through it.  Measured on the first kernel (k_render, mgpu_kernels.hip), a wave that runs
// "trace every lane's ray to completion, then shade every lane" spends 40 node steps and 40 triangle steps per ray
// batch where 5.5 and 7.7 would do (lane utilisation 14 % / 19 %): rays of very different length share a wave and
// everybody waits for the longest.  Here every lane carries an explicit state
//
//     NODE  : box tests.  BVH in LDS: pop up to six 64-byte nodes, slab-test each, push children / open a leaf.  BVH in HBM:
//             enter up to three interior nodes through their 128-byte wide records (both child boxes at once, near child
//             entered directly, far child stacked with its tmin: wide_node_step)     (bvh_accel.cc:805-834, 550-593)
//     TRI   : test the open leaf's triangles, in leaf order; with at most 32 open leaves in the wave 2 or 4 lanes share
//             each (shared_leaves_step)                                               (bvh_accel.cc:595-697)
//     SHADE : finish the ray (plane, miss / bounce logic, sampling), start the next ray, path, pass or pixel
//                                                                                (render.cc:381-456, 657-681)
//
// and each trip of the wave loop executes ONE body with exactly the lanes that wait for it active (the rule that picks it
// is at the head of the loop).  A lane that finishes its ray early gets shaded and re-armed while its neighbours are still
// traversing, so nobody waits for the longest ray any more; per-ray operation order (pop order, leaf order, RNG draws) is
// untouched, hence results are bit-identical to k_render and to the oracle.
//
// Work distribution: the unit handed to a wave is one (8x8 pixel tile, pass) pair = 64 eye paths, drawn 2 or 4 at a time
// from one of eight per-XCD counters; inside the wave, a lane whose path ends takes the next free path of the wave's
// current item at its next SHADE step.  Items are this fine because path cost varies ~50x over the frame (sky: one
// root-miss ray, Suzanne: five deep traversals): with a lane owning a pixel for all its passes the slowest wave ran
// 2.2x longer than the median one and set the frame time.  The price is that a pixel's passes are no longer summed by
// one lane, so every pass's float radiance goes to its own plane of `pass_buf` and k_accumulate_tiled adds the planes in
// pass order afterwards -- the same float32 additions, in the same order, as Render() + AccumImage
// (main_sdl.cc:138-143).  With passes == 1 the radiance is written straight into the image.
//
// Scene placement: with LDS_SCENE the whole BVH (64 B nodes + 80 B triangles) is staged once per workgroup into LDS
// (cornellbox_suzanne: 13 KB + 78 KB) next to the traversal stacks; otherwise wide records and triangles are read from HBM
// through L1/L2 and the LDS holds the far-child stacks.
#include <type_traits>
#include "mgpu_device.hpp"
#include "mgpu_kernels.hpp"

#include <climits>
#include <mutex>

#ifdef MGPU_EMU_STATS
extern "C" unsigned long long emu_stats[64];
extern "C" unsigned char *emu_log;
extern "C" unsigned long long emu_log_n, emu_log_cap;
#endif

namespace mgpu {

enum : int { ST_NODE = 0, ST_TRI = 1, ST_SHADE = 2, ST_IDLE = 3 };


// NODE runs when cN * weight >= cT.  BVH in LDS (C2): 1 -> 6.72, 2 -> 6.66, 4 -> 6.58, 8 -> 6.58, 64 -> 6.80 ms.  BVH in
// HBM: 1 -> 22.9 / 6.63 ms (teapot / 1M grid), 4 -> 23.3 / 6.91, 16 -> 24.3 / 7.44.
// Deferred path start: lanes whose path has ended take their next path only when at least MGPU_START_MIN of them ask
// (or nobody in the wave is traversing); until then they stay parked in SHADE without a ray.  The path-start body (RNG
// seeding, camera ray: ~150 instructions) then runs for more lanes at once.  Parked lanes do not count towards
// MGPU_SHADE_MIN; MGPU_START_FORCE of them trigger a SHADE step on their own.  BVH in LDS: 12 / 16 (C2 6.47 -> 6.39 ms,
// an eighth of the frame 1.27 -> 1.23; 0 / 65 switch it off).
// BVH in HBM: 8 / 12 (teapot 22.1 -> 21.7 ms, 1M grid 6.21 -> 6.08, 10M grid 84.8 -> 85.1; 12 / 16: 21.6 / 6.22 / 86.6).
#ifndef MGPU_START_MIN_LDS
#define MGPU_START_MIN_LDS 12
#define MGPU_START_FORCE_LDS 16
#endif
#ifndef MGPU_START_MIN_PRIM // with the primary rays staged in LDS a path start is a copy: 8 / 12 (5.38 ms) against 12 / 16 (5.44), 4 / 8 (5.42), 1 / 4 (5.38 .. 5.45)
#define MGPU_START_MIN_PRIM 8
#define MGPU_START_FORCE_PRIM 12
#endif
#ifndef MGPU_START_MIN_HBM
#define MGPU_START_MIN_HBM 8
#define MGPU_START_FORCE_HBM 12
#endif
#ifndef MGPU_NODE_WEIGHT_LDS
#define MGPU_NODE_WEIGHT_LDS 1
#endif
#ifndef MGPU_NODE_WEIGHT_HBM
#define MGPU_NODE_WEIGHT_HBM 1
#endif
#ifndef MGPU_TRI_WEIGHT_LDS
#define MGPU_TRI_WEIGHT_LDS 4
#endif
#ifndef MGPU_TRI_WEIGHT_HBM
#define MGPU_TRI_WEIGHT_HBM 4
#endif
#ifndef MGPU_SHARE8_MAX
#define MGPU_SHARE8_MAX 0 // open leaves up to which 8 lanes share one (0: never)
#endif
#ifndef MGPU_SHARE4_MAX
#define MGPU_SHARE4_MAX 16 // ... up to which 4 lanes share one (above: 2)
#endif
#ifndef MGPU_SHADE_MIN
#define MGPU_SHADE_MIN 36
#endif

// 4 waves per SIMD (<= 128 VGPRs; the compiler then keeps ~40 cold path-state dwords in scratch, touched only by
// SHADE): measured 12.3 -> 9.4 ms on the 1M-triangle grid and 38.7 -> 28.0 ms on teapot vs the natural 175-VGPR /
// 2-wave allocation; 5 and 6 waves spill into the NODE / TRI bodies and lose (11.1 / 15.6 ms).
// items a workgroup reserves per global-counter fetch: 16 waves share them with the BVH in LDS, 4 waves otherwise
#ifndef MGPU_WG_CHUNK_LDS
#define MGPU_WG_CHUNK_LDS 16
#endif
#ifndef MGPU_WG_CHUNK_HBM
#define MGPU_WG_CHUNK_HBM 8
#endif
#ifndef MGPU_NODES_PER_STEP
#define MGPU_NODES_PER_STEP 6
#endif
#ifndef MGPU_WIDE_PER_STEP
#define MGPU_WIDE_PER_STEP 3 // interior nodes entered per NODE step with the BVH in HBM (two box tests each)
#endif
#ifndef MGPU_SINCOS_TURN
#define MGPU_SINCOS_TURN 1
#endif
#ifndef MGPU_TAIL_TABLE
#define MGPU_TAIL_TABLE 1
#endif
#ifndef MGPU_TAIL_RECIP
#define MGPU_TAIL_RECIP 1
#endif
#ifndef MGPU_OCC
#define MGPU_OCC 0 // 1: active-lane accounting (kOcc* words); built into libmallie_mgpu_occ.so only, see mallie_amd/build.py
#endif
#ifndef MGPU_ROOT_AT_ARM
#define MGPU_ROOT_AT_ARM 0 // experiment (profiles/experiments/README.md, round 4: measured, not kept); LDS-resident scene: the root box test where the ray is armed (see there)
#endif
#ifndef MGPU_PRIM_LDS
#define MGPU_PRIM_LDS 1 // LDS-resident scene: the 64 primary rays of a work item generated by the whole wave when the item is taken (40 KB)
#endif
#ifndef MGPU_SHARED_LEAVES
#define MGPU_SHARED_LEAVES 1
#endif
#ifndef MGPU_TRIS_PER_STEP
#define MGPU_TRIS_PER_STEP 8 // trips of a TRI step (with leaf hints: 16: 5.22-5.25, 8: 5.12-5.14, 6: 5.24-5.26, 4: 5.10-5.13, 3: 5.29, 2: 5.24-5.26 ms on C2; teapot / 1 M grid 4.63 / 4.52, 4.59 / 4.50 at 8, 4.63 / 4.54 at 4)
#endif
#ifndef MGPU_SM_MIN_WAVES
#define MGPU_SM_MIN_WAVES 4
#endif
// GREY: every material of the scene has three equal diffuse channels (all the reference can load from .obj / .eson; checked
// when the scene is created).  The three channels of throughput and radiance then perform identical operations on identical
// values from the first to the last step of a path, so one is carried and the result copied: same bits, four registers less.
// SE: entry type of the LDS-resident scene's traversal stack -- a node index: one byte while the tree has at most 256 nodes,
// two up to 65 536 -- `P.stack_cap` (= tree depth + 1, what the reference's pop / push order can ever hold) of them per lane.
// (Until round 3: 16 / 24 / 32 four-byte entries, 64 KB of LDS for a 205-node tree of depth 13 that needs 14 KB.)
template <typename SE> struct LStack {
  SE *lds; // &s_stack[wave][0][lane]: entry-major, lane-minor
  __device__ __forceinline__ void put(int i, uint32_t v) const { lds[i * 64] = (SE)v; }
  __device__ __forceinline__ uint32_t get(int i) const { return (uint32_t)lds[i * 64]; }
};

// PRIM (LDS-resident scene with 40 KB of LDS to spare): see s_prim below.
template <typename SE, bool LDS_SCENE, int BLOCK, bool GREY, bool PRIM = false>
__global__ __launch_bounds__(BLOCK, (BLOCK == 1024 ? 4 : (BLOCK == 768 ? 3 : MGPU_SM_MIN_WAVES))) void k_render_sm(DScene sc, RenderParams P_arg) {
  // The launch parameters live in LDS, not in scalar registers: the traversal bodies use none of them, SHADE uses
  // nearly all, and ~60 kernel-argument SGPRs kept alive across the loop were being spilled to VGPR lanes.
  __shared__ RenderParams s_P;
  __shared__ SincosTable s_azimuth; // the cosine sampler's azimuth table (mgpu_sincos.hpp)
  if (threadIdx.x == 0) s_P = P_arg;
  if (MGPU_SINCOS_TURN) sincos_table_fill(s_azimuth, threadIdx.x, BLOCK);
  __syncthreads();
  const RenderParams &P = s_P;
  MGPU_DYN_SHARED(unsigned char, smem);
  constexpr int kWaves = BLOCK / 64;
  SE *s_stack = reinterpret_cast<SE *>(smem); // [kWaves][stack_cap][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  // traversal stack: LDS_SCENE walks the reference's 64-byte nodes staged in LDS with a stack of node indices; with the
  // BVH in HBM the wide form is used (mgpu_device.hpp, wide_node_step): far children with their tmin
  const uint32_t stack_cap = LDS_SCENE ? P.stack_cap : 0u;
  LStack<SE> stk;
  stk.lds = s_stack + ((size_t)wave * stack_cap) * 64 + lane;
  using WS = WStack<kWideStackLds>;
  WS wstk;
  if (!LDS_SCENE) {
    wstk.bind(smem + (size_t)wave * WS::kWaveBytes, lane);
    wstk.overflow = sc.wstack_overflow ? sc.wstack_overflow + gid * sc.woverflow_cap : nullptr;
  }

  // ---- BVH in HBM, 1024-thread workgroup: the treelet (mgpu_device.hpp, kWTreelet) behind the far-child stacks ----
  constexpr bool TL = !LDS_SCENE && BLOCK >= 768;
  const unsigned char *lds_treelet = smem + (size_t)kWaves * WS::kWaveBytes;
  if (TL) {
    const uint4 *src = reinterpret_cast<const uint4 *>(sc.treelet);
    uint4 *dst = reinterpret_cast<uint4 *>(const_cast<unsigned char *>(lds_treelet));
    const uint32_t n16 = P.lds_nodes_bytes >> 4;
    for (uint32_t i = threadIdx.x; i < n16; i += BLOCK) dst[i] = src[i];
    __syncthreads();
  }
  // ---- optional: stage nodes + triangles into LDS -------------------------------------------------------------
  const unsigned char *lds_nodes = smem + lds_stack_bytes(kWaves, stack_cap, sizeof(SE));
  const unsigned char *lds_tris = lds_nodes + (size_t)P.lds_nodes_bytes;
  if (LDS_SCENE) {
    const uint4 *src = reinterpret_cast<const uint4 *>(sc.nodes);
    uint4 *dst = reinterpret_cast<uint4 *>(const_cast<unsigned char *>(lds_nodes));
    const uint32_t n16 = P.lds_nodes_bytes >> 4;
    for (uint32_t i = threadIdx.x; i < n16; i += BLOCK) dst[i] = src[i];
    const uint4 *src2 = reinterpret_cast<const uint4 *>(sc.tris);
    uint4 *dst2 = reinterpret_cast<uint4 *>(const_cast<unsigned char *>(lds_tris));
    const uint32_t t16 = P.lds_tris_bytes >> 4;
    for (uint32_t i = threadIdx.x; i < t16; i += BLOCK) dst2[i] = src2[i];
    __syncthreads();
  }

  // PRIM: a (tile, pass) item's 64 primary rays -- start state, jitter, camera direction -- are made by ALL lanes of the wave at the
  // moment the item is taken and parked in LDS (40 bytes each); a lane that is handed a path copies its ray from there.  The
  // path-start body (~170 instructions: two splitmix64, two draws, a normalisation) then runs once per 64 paths with 64 lanes
  // instead of in two SHADE steps out of three with whoever asks.
  // HBM-resident scene (round 6): the LDS is full (stacks + treelet), so the same 40-byte slots live in device memory, 64 per wave of the
  // launch (RenderParams::prim_stage); only the wave that wrote them reads them, a few instructions later, so they stay in the CU's L1.
  // Measured on the ISA interpreter (tools/isa_profile.py): the path-start body ran for 38 lanes on average, now always for 64.
  struct PrimRay {
    double d[3];
    uint32_t s[4];
  };
  PrimRay *s_prim = LDS_SCENE ? reinterpret_cast<PrimRay *>(const_cast<unsigned char *>(lds_tris) + (((size_t)P.lds_tris_bytes + 15) & ~(size_t)15)) + (size_t)wave * 64
                              : (PRIM ? reinterpret_cast<PrimRay *>(P.prim_stage) + ((size_t)blockIdx.x * kWaves + (size_t)wave) * 64 : nullptr);
  // Leaf hints (mgpu_device.hpp, leaf_hint_make).  LDS-resident scene: P.lds_hint_cap records of 96 bytes behind the staging above,
  // made here from the LDS copy of the triangles; a leaf's axis field (unused by the reference's traversal) becomes (hint + 1) << 16,
  // which the NODE step adds to tri_end when it opens the leaf (slots of an LDS-resident scene stay below 2^16).  (For the
  // HBM-resident scene -- records made when the scene is created, found through packed leaf tags -- the record is one more
  // dependent fetch in front of every TRI step: built, exact, 2-4 % slower; profiles/experiments/leaf_hints_hbm.diff.txt.)
  constexpr uint32_t kHintMinTris = MGPU_HINT_MIN;
  unsigned char *lds_hints = const_cast<unsigned char *>(lds_tris) + (((size_t)P.lds_tris_bytes + 15) & ~(size_t)15) + (PRIM ? (size_t)kWaves * 64 * sizeof(PrimRay) : 0);
  if (LDS_SCENE && MGPU_LEAF_HINTS) {
    __shared__ uint32_t s_nhints;
    if (threadIdx.x == 0) s_nhints = 0u;
    __syncthreads();
    const uint32_t nn_lds = P.lds_nodes_bytes >> 6;
    for (uint32_t i = threadIdx.x; i < nn_lds; i += BLOCK) {
      uint32_t *nd = reinterpret_cast<uint32_t *>(const_cast<unsigned char *>(lds_nodes) + (size_t)i * 64);
      if (nd[12] == 0u) continue; // interior node: its axis field is the split axis
      uint32_t code = 0u;
      const uint32_t n = nd[14], first = nd[15];
      if (P.lds_hint_cap != 0u && n >= kHintMinTris && n <= 64u) {
        float rec[kHintFloats];
        if (leaf_hint_make([&](uint32_t k) { return reinterpret_cast<const double *>(lds_tris + (size_t)(first + k) * 80); }, n,
                           (double)MGPU_HINT_WORTH, P.hint_c, P.hint_q, rec)) {
          const uint32_t h = atomicAdd(&s_nhints, 1u);
          if (h < P.lds_hint_cap) {
            leaf_hint_pack(rec, reinterpret_cast<float *>(lds_hints + (size_t)h * (kHintFloats * 4)));
            code = (h + 1u) << 16;
          }
        }
      }
      nd[13] = code;
    }
    __syncthreads();
  }
  const int win_w = P.x1 - P.x0;
  const uint32_t tiles_x = (uint32_t)(win_w + 7) >> 3;
  const uint32_t tiles_y = (uint32_t)(P.n_rows + 7) >> 3;
  const uint32_t total_tiles = tiles_x * tiles_y;

  // Work cursor.  The unit handed to a wave is one item = (8x8 tile, pass) = 64 eye paths; `in_item` is the wave's
  // position inside its current item (wave-uniform).  Items come from a two-level counter: the workgroup reserves
  // kWgChunk items at a time from one of kShards global counters (one per XCD: less contention on each, a workgroup
  // draws from its own XCD's counter first) into an LDS cursor, and its waves take single items from that cursor with
  // one LDS atomic.  Compared with every wave reserving its own chunk this needs ~10x fewer global atomics (thousands
  // of waves crossing a cheap region used to queue up on the counters) and it leaves less reserved-but-unstarted work
  // per CU when the counters run dry, i.e. a more even end of the launch.
  // HBM-resident scenes: part s = the s-th CONTIGUOUS eighth of the items, so the waves of one XCD (one L2) walk one
  // image region.  LDS-resident scenes have no L2 locality to protect: part s = every 8th item (item = k * kShards + s),
  // which balances the parts by construction.
  const uint32_t total_items = total_tiles * (uint32_t)P.passes;
  uint32_t in_item = 64;
  bool exhausted = false;
  constexpr uint32_t kWgChunk = LDS_SCENE ? (uint32_t)MGPU_WG_CHUNK_LDS : (uint32_t)MGPU_WG_CHUNK_HBM * (BLOCK >= 768 ? 2u : 1u);
  const uint32_t shard_items = (total_items + (uint32_t)kShards - 1) / (uint32_t)kShards;
  uint32_t home_shard = 0;
  uint32_t item_tile = 0, item_pass = 0; // wave-uniform: tile and pass of the current item
  MGPU_XCC_ID(home_shard); // which XCD this workgroup runs on (0..7)
  home_shard &= 7u;
  // LDS cursor: hi32 = end, lo32 = next, both (shard << 28) | index of the item inside its shard's part
  __shared__ uint32_t s_occ[8]; // occupancy accounting of the workgroup: kOccNodeTrips.. in that order
  __shared__ unsigned char s_owner[BLOCK]; // TRI step with shared leaves: lane of the k-th open leaf, per wave
  __shared__ unsigned long long wg_cursor;
  __shared__ uint32_t wg_lock, wg_shard_off, wg_dry;
#if MGPU_OCC
  if (threadIdx.x < 8) s_occ[threadIdx.x] = 0u;
#endif
  if (threadIdx.x == 0) {
    wg_cursor = 0ull;
    wg_lock = 0u;
    wg_shard_off = 0u;
    wg_dry = 0u;
  }
  __syncthreads();

  // ---- per-lane path state --------------------------------------------------------------------------------------
  int st = ST_SHADE;       // everybody starts by asking for work
  bool have_ray = false;   // a traversal result is waiting to be shaded
  bool have_path = false;  // lane was handed a fresh (pixel, pass) it has not started yet
  uint32_t lx = 0, ly = 0;
  int pass = 0;
  Rng rng{1, 0, 0, 0};
  V3 org = v3(0, 0, 0), dir = v3(0, 0, 1);
  double thr0 = 1, thr1 = 1, thr2 = 1; // the radiance is not path state: it stays 0 until the step that ends the path (below)
  int pathLength = 1;
  uint32_t last_mat = kNoMaterial;
  uint32_t cost_base = 0;  // n_nodes + n_tris + 16 * n_rays when the current path started
  // ---- per-lane traversal state ---------------------------------------------------------------------------------
  double ix = 0, iy = 0, iz = 0;
  uint32_t sgn = 0; // bit k: dir[k] < 0 (dirSign, bvh_accel.cc:786-790)
  bool ray_plain = false; // this ray may take the min/max form of the slab test (see the NODE step)
  bool ray_hints = false; // ... and may consult leaf hints (mgpu_device.hpp, leaf_hint_make: its origin is within the launch's reach)
  int sp = -1;           // LDS_SCENE: index of the stack top; wide form: number of far children on the stack
  uint32_t cur = kWNone; // wide form: record to enter next (kWNone: pop)
  double bt = kDblMax, bu = 0, bv = 0;
  uint32_t bslot = kNoHit;
  uint32_t tri_cur = 0, tri_end = 0;
  // ---- counters -------------------------------------------------------------------------------------------------
  uint32_t n_rays = 0, n_nodes = 0, n_tris = 0, trace_calls = 0, paths = 0;
  // Occupancy accounting (kOcc* in mgpu_kernels.hpp): a step is booked when the low bits of the shader clock say so (one
  // step in kSampleEvery on average, independent of what the step does), into LDS words of the workgroup -- nothing is
  // kept in registers between steps.  Trips of a body's loop = the largest per-lane iteration count `d`, lanes over those
  // trips = the sum of the counts; both from one ballot per possible count.
#if MGPU_OCC
  auto occ_sampled = [&]() -> bool { return (__builtin_amdgcn_s_memtime() & (unsigned long long)((kSampleEvery - 1) << 2)) == 0ull; };
#else
  auto occ_sampled = [&]() -> bool { return false; }; // product build: the accounting folds away (it costs 3-4 % in registers)
#endif
  auto occ_book = [&](uint32_t d, int max_d, int word) {
    uint32_t trips = 0, lanes = 0;
    for (int k = 1; k <= max_d; ++k) {
      const unsigned long long b = MGPU_BALLOT(d >= (uint32_t)k);
      if (b == 0ull) break;
      trips += 1u;
      lanes += (uint32_t)__popcll(b);
    }
    if (lane == 0) {
      atomicAdd(&s_occ[word], trips);
      atomicAdd(&s_occ[word + 1], lanes);
      atomicAdd(&s_occ[6 + (word >> 1)], 1u); // steps booked
    }
  };
  bool probe_on = false;
#ifdef MGPU_UTIL
  uint32_t u_node = 0, u_tri = 0, u_shade = 0, u_shade_lanes = 0;
  unsigned long long u_hist = 0;
  uint32_t u_tail_steps = 0, u_bounce_steps = 0, u_start_steps = 0; // SHADE steps in which the sub-body ran (booked by its first lane)
  uint32_t u_node_it = 0, u_tri_it = 0; // loop iterations inside NODE / TRI steps (one lane of the wave books each)
  unsigned long long u_hc[4] = {0, 0, 0, 0};
  uint32_t u_hint_fresh = 0, u_hint_dropped = 0, u_hint_empty = 0, u_hint_steps = 0; // leaf hints: consulted, triangles dropped, leaves dropped whole, TRI steps with a consultation
  unsigned long long cyc_node = 0, cyc_tri = 0, cyc_shade = 0, cyc_t0 = 0, cyc_s = 0;
  unsigned long long cyc_sub[6] = {0, 0, 0, 0, 0, 0};
#define MGPU_TICK() (cyc_t0 = clock64())
#define MGPU_TOCK(acc) (acc += clock64() - cyc_t0)
#else
#define MGPU_TICK()
#define MGPU_TOCK(acc)
#endif

#ifdef MGPU_UTIL
  const unsigned long long cyc_loop0 = clock64();
  const unsigned long long wall_loop0 = wall_clock64(); // 100 MHz, one base for the whole device (clock64 has many)
  unsigned long long cyc_dry = 0; // when this wave first found the work cursor exhausted
  bool dry_mark = false;
  uint32_t dry_rays = 0, dry_steps = 0, dry_active = 0, dry_plen = 0, dry_inpath = 0, steps_n = 0, steps_t = 0, steps_s = 0;
#endif
  for (;;) {
    const unsigned long long mN = MGPU_BALLOT(st == ST_NODE);
    const unsigned long long mT0 = MGPU_BALLOT(st == ST_TRI);
    const unsigned long long mS = MGPU_BALLOT(st == ST_SHADE);
    const int cN = __popcll(mN), cT0 = __popcll(mT0), cS = __popcll(mS);
    if ((cN | cT0 | cS) == 0) break;
#ifdef MGPU_UTIL
    if (cyc_dry) ++dry_steps;
#endif

    // Scheduling rule: SHADE is by far the most expensive body (fp64 sqrt/div/sincos), so it runs only when at least
    // MGPU_SHADE_MIN lanes have a ray to finish, or MGPU_START_FORCE lanes are parked between paths, or nothing else is
    // runnable; otherwise NODE runs unless TRI has several times more lanes waiting (MGPU_NODE_WEIGHT_*).
    const int cReal = __popcll(MGPU_BALLOT(st == ST_SHADE && have_ray)); // lanes with a ray to finish (not parked between paths)
    const bool run_shade = (cReal >= MGPU_SHADE_MIN) || (cN == 0 && cT0 == 0) || (cS - cReal >= (PRIM ? MGPU_START_FORCE_PRIM : (LDS_SCENE ? MGPU_START_FORCE_LDS : MGPU_START_FORCE_HBM)));
    if (!run_shade && cN * (LDS_SCENE ? MGPU_NODE_WEIGHT_LDS : MGPU_NODE_WEIGHT_HBM) >= cT0 * (LDS_SCENE ? MGPU_TRI_WEIGHT_LDS : MGPU_TRI_WEIGHT_HBM)) {
      // ================================ NODE step ================================
      MGPU_TICK();
      const bool all_plain = MGPU_BALLOT(st == ST_NODE && !ray_plain) == 0ull; // wave-uniform
      const bool occ_sample = occ_sampled();
      const uint32_t occ_n0 = MGPU_OCC ? n_nodes : 0u;
#ifdef MGPU_EMU_STATS
      const uint32_t emu_n0 = n_nodes;
      const bool st_was_node = st == ST_NODE;
#endif
      if (st == ST_NODE) {
        const bool sx = (sgn & 1u) != 0u, sy = (sgn & 2u) != 0u, sz = (sgn & 4u) != 0u;
#ifdef MGPU_UTIL
        if (lane == __ffsll((long long)mN) - 1) u_node++;
#endif
        // Two bodies of the same loop: slab_hit<true> (min/max form, 14 VALU instructions fewer per box) when every lane's
        // ray qualifies, the literal form for the step otherwise (mgpu_device.hpp).
        auto node_pops = [&](auto plain_tag) {
          constexpr bool kPlain = decltype(plain_tag)::value;
#pragma unroll 1
          for (int rep = 0; rep < MGPU_NODES_PER_STEP; ++rep) {
#ifdef MGPU_UTIL
            if (lane == __ffsll((long long)MGPU_BALLOT(1)) - 1) u_node_it++;
#endif
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
            const bool hit = slab_hit<kPlain>(b0, b1, b2, org, ix, iy, iz, sx, sy, sz, bt);
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
                if (LDS_SCENE && MGPU_LEAF_HINTS) tri_end += (uint32_t)meta.y; // (hint + 1) << 16 or 0: taken off when the TRI work starts
                st = ST_TRI;
              }
            }
            if (st != ST_NODE || sp < 0) break;
          }
        };
        if constexpr (LDS_SCENE) {
          if (all_plain) node_pops(std::true_type{});
          else node_pops(std::false_type{});
          if (st == ST_NODE && sp < 0) st = ST_SHADE;
        } else {
          int r;
#ifdef MGPU_EXP_ONE_LOOP
          if (true)
            r = wide_node_step<true, MGPU_WIDE_PER_STEP, kWideStackLds, TL>(sc.wnodes, wstk, org, ix, iy, iz, sx, sy, sz, sgn, bt, cur, sp,
                                                                           tri_cur, tri_end, n_nodes, lds_treelet, all_plain);
          else
#endif
          if (all_plain)
            r = wide_node_step<true, MGPU_WIDE_PER_STEP, kWideStackLds, TL>(sc.wnodes, wstk, org, ix, iy, iz, sx, sy, sz, sgn, bt, cur, sp,
                                                                           tri_cur, tri_end, n_nodes, lds_treelet);
          else
            r = wide_node_step<false, MGPU_WIDE_PER_STEP, kWideStackLds, TL>(sc.wnodes, wstk, org, ix, iy, iz, sx, sy, sz, sgn, bt, cur, sp,
                                                                            tri_cur, tri_end, n_nodes, lds_treelet);
          if (r == WT_TRI) st = ST_TRI;
          else if (r == WT_DONE) st = ST_SHADE;
        }
      }
      if (occ_sample) occ_book(LDS_SCENE ? n_nodes - occ_n0 : (n_nodes - occ_n0) >> 1, LDS_SCENE ? MGPU_NODES_PER_STEP : MGPU_WIDE_PER_STEP,
                               0);
#ifdef MGPU_EMU_STATS // (emulator builds only: NODE steps, the wave's trips of the repetition loop, lane-trips)
      {
        const uint32_t reps = (st_was_node ? (LDS_SCENE ? n_nodes - emu_n0 : (n_nodes - emu_n0 + 1u) >> 1) : 0u);
        uint32_t mx = reps, sum = reps;
        for (int x = 1; x < 64; x <<= 1) {
          mx = max(mx, (uint32_t)__shfl_xor((int)mx, x));
          sum += (uint32_t)__shfl_xor((int)sum, x);
        }
        if (lane == 0) {
          atomicAdd(&emu_stats[45], 1ull);
          atomicAdd(&emu_stats[46], (unsigned long long)mx);
          atomicAdd(&emu_stats[47], (unsigned long long)sum);
          atomicAdd(&emu_stats[48], (unsigned long long)cN);
        }
      }
#endif
#ifdef MGPU_UTIL
      if (cyc_dry) ++steps_n;
#endif
      MGPU_TOCK(cyc_node);
    } else if (!run_shade) {
      // ================================ TRI step =================================
      MGPU_TICK();
      bool shared_done = false;
      unsigned long long mT = mT0;
      int cT = cT0;
      if (LDS_SCENE && MGPU_LEAF_HINTS) { // leaves opened since the last TRI step: their hints (mgpu_device.hpp, leaf_hint_make)
        const bool fresh = st == ST_TRI && (tri_end >> 16) != 0u;
        if (MGPU_BALLOT(fresh) != 0ull) {
          if (fresh) {
            const float *hp = reinterpret_cast<const float *>(lds_hints + (size_t)((tri_end >> 16) - 1u) * (kHintFloats * 4));
            tri_end &= 0xFFFFu;
            if (ray_hints) {
              const float4 f0 = *reinterpret_cast<const float4 *>(hp), f1 = *reinterpret_cast<const float4 *>(hp + 4),
                           f2 = *reinterpret_cast<const float4 *>(hp + 8), cA = *reinterpret_cast<const float4 *>(hp + 12),
                           cB = *reinterpret_cast<const float4 *>(hp + 16);
              const uint32_t m = __float_as_uint(hp[20]);
              const uint32_t dropped = leaf_hint_apply(f0, f1, f2, cA, cB, m, org, dir, ix, iy, iz, bt, tri_cur, tri_end);
              n_tris += dropped; // the tests the reference makes on the dropped part
#ifdef MGPU_EMU_STATS // (emulator builds only: how many consultations, how many tests they drop, how many leaves go whole)
              atomicAdd(&emu_stats[40], 1ull);
              atomicAdd(&emu_stats[41], (unsigned long long)dropped);
              if (tri_cur == tri_end) atomicAdd(&emu_stats[42], 1ull);
              atomicAdd(&emu_stats[43], (unsigned long long)(pathLength == 1 ? 1 : 0));
              if (pathLength == 1) atomicAdd(&emu_stats[44], (unsigned long long)dropped);
#endif
#ifdef MGPU_UTIL
              u_hint_fresh++;
              u_hint_dropped += dropped;
#ifdef MGPU_UTIL_HINTCLASS
              {
                const double cx = 0.5 * ((double)f0.x + (double)f0.w) - org.x, cy = 0.5 * ((double)f0.y + (double)f1.x) - org.y, cz = 0.5 * ((double)f0.z + (double)f1.y) - org.z;
                const double d2 = cx * cx + cy * cy + cz * cz;
                const int cls = pathLength == 1 ? 0 : (d2 < 2.25 ? 1 : (d2 < 25.0 ? 2 : 3));
                u_hc[cls] += 1ull | ((unsigned long long)dropped << 32);
              }
#endif
              if (tri_cur == tri_end) u_hint_empty++;
#endif
            }
            if (tri_cur == tri_end) st = sp < 0 ? ST_SHADE : ST_NODE; // nothing left of the leaf
          }
#ifdef MGPU_UTIL
          if (lane == __ffsll((long long)MGPU_BALLOT(1)) - 1) u_hint_steps++;
#endif
          mT = MGPU_BALLOT(st == ST_TRI);
          cT = __popcll(mT);
        }
      }
      const bool occ_sample = occ_sampled();
      const uint32_t occ_t0 = MGPU_OCC ? tri_cur : 0u;
#ifdef MGPU_EMU_STATS // (emulator builds only: the shape of the TRI steps -- open leaves, tests due, the longest run -- for costing hand-out schemes)
      if (cT != 0) {
        const uint32_t len = st == ST_TRI ? tri_end - tri_cur : 0u;
        uint32_t mx = len, sum = len;
        for (int x = 1; x < 64; x <<= 1) {
          mx = max(mx, (uint32_t)__shfl_xor((int)mx, x));
          sum += (uint32_t)__shfl_xor((int)sum, x);
        }
        if (lane == 0) {
          const uint32_t m = cT <= 16 ? 4u : (cT <= 32 ? 2u : 1u), cap = (uint32_t)MGPU_TRIS_PER_STEP;
          const uint32_t trips_now = min(cap, (mx + m - 1u) / m);
          // tests this step performs: every run advances by up to cap * m
          emu_stats[0] += 1;                       // TRI steps
          emu_stats[1] += (unsigned long long)cT;  // open leaves
          emu_stats[2] += sum;                     // tests due in the open leaves
          emu_stats[3] += trips_now;               // trips of the step as built
          emu_stats[4] += (sum + 63u) / 64u;       // trips if the (leaf, triangle) pairs were dealt to the lanes one by one, whole runs
          emu_stats[5] += mx;
          emu_stats[8 + min((uint32_t)cT, 64u) / 4u] += 1; // steps by open leaves / 4
        }
        if (emu_log && emu_log_n + 66 < emu_log_cap) { // per-step record: cT, then every open leaf's run length (lane order), 0-terminated
          unsigned long long base = 0;
          if (lane == 0) base = atomicAdd(&emu_log_n, (unsigned long long)(cT + 2)); // (the waves of a workgroup take turns: reserve, then fill)
          base = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(base >> 32), 0) << 32) | (unsigned long long)(uint32_t)__shfl((int)(uint32_t)base, 0);
          if (lane == 0) {
            emu_log[base] = (unsigned char)cT;
            emu_log[base + 1 + cT] = 0;
          }
          if (st == ST_TRI) emu_log[base + 1 + __popcll(mT & ((1ull << lane) - 1ull))] = (unsigned char)min(len, 255u);
        }
        // tests done by the step as built
        if (st == ST_TRI) atomicAdd(&emu_stats[6], (unsigned long long)min(len, (cT <= 32 ? (uint32_t)(MGPU_TRIS_PER_STEP) * (cT <= 16 ? 4u : 2u) : (uint32_t)MGPU_TRIS_PER_STEP)));
      }
#endif
#if MGPU_SHARED_LEAVES
      if (cT != 0 && cT <= 32) { // mgpu_device.hpp, shared_leaves_step: 2, 4 (or 8) lanes per open leaf
        uint32_t my_trips = 0;
        shared_done = shared_leaves_step<LDS_SCENE, MGPU_TRIS_PER_STEP>(mT, cT, cT <= MGPU_SHARE8_MAX ? 3 : (cT <= MGPU_SHARE4_MAX ? 2 : 1), lane,
                                                                       s_owner + wave * 64, st == ST_TRI, lds_tris, sc.tris, org, dir,
                                                                       tri_cur, tri_end, bt, bu, bv, bslot, n_tris, my_trips);
        if (shared_done && occ_sample) occ_book(my_trips, MGPU_TRIS_PER_STEP, 2);
      }
#endif
      if (!shared_done && st == ST_TRI) {
#ifdef MGPU_UTIL
        if (lane == __ffsll((long long)mT) - 1) u_tri++;
#endif
        // TriangleIsect, bvh_accel.cc:595-638, on the triangle in (a0..a3, e2z)
        auto tri_test = [&](double2 a0, double2 a1, double2 a2, double2 a3, double e2z) {
          ++n_tris;
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
        };
        if constexpr (LDS_SCENE) {
#pragma unroll 1
          for (int rep = 0; rep < MGPU_TRIS_PER_STEP; ++rep) {
#ifdef MGPU_UTIL
            if (lane == __ffsll((long long)MGPU_BALLOT(1)) - 1) u_tri_it++;
#endif
            const unsigned char *tp = lds_tris + (size_t)tri_cur * 80;
            const double2 a0 = *reinterpret_cast<const double2 *>(tp);
            const double2 a1 = *reinterpret_cast<const double2 *>(tp + 16);
            const double2 a2 = *reinterpret_cast<const double2 *>(tp + 32);
            const double2 a3 = *reinterpret_cast<const double2 *>(tp + 48);
            const double e2z = *reinterpret_cast<const double *>(tp + 64);
            tri_test(a0, a1, a2, a3, e2z);
            ++tri_cur;
            if (tri_cur == tri_end) break;
          }
        } else {
          // (requesting triangle i + 1 before triangle i is tested costs 38 more spilled registers and loses 10 %:
          // profiles/experiments/README.md)
#pragma unroll 1
          for (int rep = 0; rep < MGPU_TRIS_PER_STEP; ++rep) {
#ifdef MGPU_UTIL
            if (lane == __ffsll((long long)MGPU_BALLOT(1)) - 1) u_tri_it++;
#endif
            const DTri *tp = sc.tris + tri_cur;
            const double2 a0 = reinterpret_cast<const double2 *>(tp)[0], a1 = reinterpret_cast<const double2 *>(tp)[1],
                          a2 = reinterpret_cast<const double2 *>(tp)[2], a3 = reinterpret_cast<const double2 *>(tp)[3];
            const double e2z = tp->e2[2];
            tri_test(a0, a1, a2, a3, e2z);
            ++tri_cur;
            if (tri_cur == tri_end) break;
          }
        }
      }
      if (occ_sample && !shared_done) occ_book(tri_cur - occ_t0, MGPU_TRIS_PER_STEP, 2);
      if (st == ST_TRI && tri_cur == tri_end) st = (LDS_SCENE ? sp < 0 : sp == 0) ? ST_SHADE : ST_NODE;
#ifdef MGPU_UTIL
      if (cyc_dry) ++steps_t;
#endif
      MGPU_TOCK(cyc_tri);
    } else {
      MGPU_TICK();
      // ================================ SHADE step ===============================
      // Three parts: (1) lanes in SHADE finish their ray; (2) ALL lanes of the wave run the pixel hand-out so the
      // work cursor stays wave-uniform; (3) lanes in SHADE start their next path / arm their next traversal.
      const bool shade_lane = (st == ST_SHADE);
#ifdef MGPU_EMU_STATS // (emulator builds only: SHADE steps, their lanes, and how often each sub-body has a lane: miss tail, bounce, path start, arming)
      {
        const bool fin = shade_lane && have_ray;
        const bool hit_now = fin && bt < kDblMax;
        if (lane == 0) {
          atomicAdd(&emu_stats[50], 1ull);
          atomicAdd(&emu_stats[51], (unsigned long long)cS);
          atomicAdd(&emu_stats[52], (unsigned long long)cReal);
        }
        (void)hit_now;
      }
#endif
#if MGPU_OCC
      if (occ_sampled() && lane == 0) {
        atomicAdd(&s_occ[4], 1u);
        atomicAdd(&s_occ[5], (uint32_t)cS);
      }
#endif
      bool path_done = false, want_pixel = false;
      if (shade_lane) {
#ifdef MGPU_UTIL
        if (lane == __ffsll((long long)mS) - 1) {
          u_shade++;
          u_shade_lanes += (uint32_t)cS;
          // SHADE steps by lane count: < 16, 16..35 (both only when nothing else was runnable), 36..47, >= 48
          u_hist += 1ull << (16 * (cS < 16 ? 0 : cS < 36 ? 1 : cS < 48 ? 2 : 3)); // four 16-bit counters per lane
        }
#endif
#ifdef MGPU_UTIL
        cyc_s = clock64();
#endif
        path_done = !have_ray; // a lane without a ray is between paths
        if (have_ray) {
          // ---- the rest of one PathTrace loop iteration (render.cc:403-452) ----
          bool hit = bt < kDblMax; // bvh_accel.cc:838
          double t = bt;
          V3 n = v3(0, 0, 0);
          if (bslot != kNoHit) // written by TestLeafNode on every accepted triangle
            last_mat = LDS_SCENE ? *reinterpret_cast<const uint32_t *>(lds_tris + (size_t)bslot * 80 + 76) : sc.tris[bslot].mat;
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
          if (P.probe && probe_on) {
            double *rec = P.probe + (size_t)(pathLength - 1) * kProbeStride;
            rec[0] = org.x; rec[1] = org.y; rec[2] = org.z; rec[3] = dir.x; rec[4] = dir.y; rec[5] = dir.z;
            rec[6] = t; rec[7] = hit ? 1.0 : 0.0; rec[8] = (bt < kDblMax && t == bt) ? (double)bslot : -1.0;
            rec[9] = n.x; rec[10] = n.y; rec[11] = n.z; rec[12] = (double)last_mat; rec[13] = (double)pathLength;
            rec[14] = thr0; rec[15] = 0.0; // radiance before the iteration's update: nothing is added before the first miss
          }
#ifdef MGPU_UTIL
          cyc_sub[0] += clock64() - cyc_s; cyc_s = clock64();
#endif
          // PathTrace adds to the radiance only on a miss (render.cc:409-418), and the first miss ends the path here (its
          // continuation is evaluated in closed loop): the radiance lives in this step only, not in registers between steps
          double rad0 = 0.0, rad1 = 0.0, rad2 = 0.0;
#ifdef MGPU_EMU_STATS
          atomicAdd(&emu_stats[53], 1ull);                                                    // rays finished
          if (!hit) atomicAdd(&emu_stats[pathLength < 2 ? 54 : 55], 1ull);                   // primary misses / misses with a tail
          else if (pathLength < P.maxPathLength) atomicAdd(&emu_stats[56], 1ull);            // bounces
#endif
          if (!hit) {
            path_done = true;
            if (pathLength < 2) {
              trace_calls += 1; // eye ray -> background: radiance stays 0 (render.cc:409-412)
            } else {
              // First miss of a path that has bounced: the reference iterates on to kMaxPathLength with the stale
              // intersection record; every one of those rays starts ~1e308 away and misses, adds
              // throughput*0.5/length and re-applies the stale material.  Their RNG draws cannot reach
              // this pixel's value, so the tail is evaluated in closed loop: same adds, same multiplies, same order.
              trace_calls += (uint32_t)P.maxPathLength;
#ifdef MGPU_UTIL
              const unsigned long long cyc_tail0 = clock64();
              const bool tail_first = lane == __ffsll((long long)MGPU_BALLOT(1)) - 1;
#endif
              double d0 = 0.5, d1 = 0.5, d2 = 0.5; // Material().diffuse default (material.h:12-15)
              const bool mul = last_mat != kNoMaterial;
              if (mul && (size_t)(int)last_mat < (size_t)sc.nm) {
                d0 = sc.mat_diffuse[3 * (size_t)last_mat + 0];
                d1 = sc.mat_diffuse[3 * (size_t)last_mat + 1];
                d2 = sc.mat_diffuse[3 * (size_t)last_mat + 2];
              }
              if (GREY || (thr0 == thr1 && thr1 == thr2 && d0 == d1 && d1 == d2)) {
                // grey path (every material the reference can load from .obj/.eson is grey): the three channels
                // perform identical operations on identical values, so evaluate one and copy -- same bits, 1/3 of the
                // fp64 divisions
                // Throughput a power of two and a 0.5-grey (or no) multiplier -- every scene the reference loads without an .mtl:
                // Material() is 0.5 grey (material.h:12-15) -- make every operand of the loop below a power-of-two multiple of
                // what it is for throughput 1, and scaling by a power of two commutes with every rounding in it (no underflow
                // above 2^-900): the sum is throughput x (the loop's result for throughput 1), which the host tabulates per
                // (multiplier on / off, first length) with the loop's own operations (RenderParams::tail_unit).
                const unsigned long long thr_bits = (unsigned long long)__double_as_longlong(thr0);
                const bool unit_ok = MGPU_TAIL_TABLE && P.maxPathLength <= 16 && d0 == 0.5 && (thr_bits & 0x000FFFFFFFFFFFFFull) == 0ull &&
                                     thr0 >= 0x1p-900 && thr0 <= 1.0;
                if (MGPU_TAIL_TABLE && !MGPU_ANY(!unit_ok)) { // (both forms give the same bits: the table when every lane's operands qualify)
                  rad0 = thr0 * P.tail_unit[mul ? 1 : 0][pathLength];
                } else if (MGPU_TAIL_RECIP && P.maxPathLength <= 16) { // x / L through the rounded reciprocal (mgpu_kernels.hpp, inv_len)
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
                rad1 = rad2 = rad0;
                if (!GREY) thr1 = thr2 = thr0;
              } else {
                const bool recip = MGPU_TAIL_RECIP && P.maxPathLength <= 16;
                for (int L = pathLength;; ++L) {
                  const double dl = (double)(unsigned)L;
                  if (recip) {
                    const double y = P.inv_len[L];
                    const double x0 = thr0 * 0.5, x1 = thr1 * 0.5, x2 = thr2 * 0.5;
                    const double q0 = x0 * y, q1 = x1 * y, q2 = x2 * y;
                    rad0 += fma(fma(-q0, dl, x0), y, q0);
                    rad1 += fma(fma(-q1, dl, x1), y, q1);
                    rad2 += fma(fma(-q2, dl, x2), y, q2);
                  } else {
                    rad0 += thr0 * 0.5 / dl;
                    rad1 += thr1 * 0.5 / dl;
                    rad2 += thr2 * 0.5 / dl;
                  }
                  if (L >= P.maxPathLength) break;
                  if (mul) { thr0 *= d0; thr1 *= d1; thr2 *= d2; }
                }
              }
#ifdef MGPU_UTIL
              if (tail_first) { cyc_sub[1] += clock64() - cyc_tail0; u_tail_steps++; }
#endif
            }
          } else if (pathLength >= P.maxPathLength) {
            path_done = true;
            trace_calls += (uint32_t)P.maxPathLength;
          } else {
#ifdef MGPU_UTIL
            const unsigned long long cyc_b0 = clock64();
            const bool bounce_first = lane == __ffsll((long long)MGPU_BALLOT(1)) - 1;
#endif
            const V3 hitP = org + scale(dir, t);
            (void)rng_next(rng); // `double r = randomreal();` drawn and never used (render.cc:430)
            const double ndoti = dot(n, neg(dir));
            if (ndoti < 0.0) n = neg(n);
            const V3 sd = sample_diffuse_t<MGPU_SINCOS_TURN != 0>(n, rng, &s_azimuth);
            if (last_mat != kNoMaterial) { // Scene::GetMaterial, scene.h:58-65
              if ((size_t)(int)last_mat < (size_t)sc.nm) {
                thr0 *= sc.mat_diffuse[3 * (size_t)last_mat + 0];
                if (!GREY) {
                  thr1 *= sc.mat_diffuse[3 * (size_t)last_mat + 1];
                  thr2 *= sc.mat_diffuse[3 * (size_t)last_mat + 2];
                }
              } else {
                thr0 *= 0.5;
                if (!GREY) { thr1 *= 0.5; thr2 *= 0.5; }
              }
            }
            org = hitP + scale(sd, 1.0e-3);
            dir = sd;
            ++pathLength;
#ifdef MGPU_UTIL
            if (bounce_first) { cyc_sub[3] += clock64() - cyc_b0; u_bounce_steps++; }
#endif
          }
          if (path_done) {
            // image[...] = radiance (double -> float, render.cc:673-675); passes are summed later, in order
            // Several passes: planes are TILE-major (a tile's 64 pixels = 768 contiguous bytes = six whole 128-byte lines that
            // only this (tile, pass) item writes; in image order a tile row is 96 bytes straddling lines shared with the
            // neighbouring tiles, i.e. with other waves on other XCDs, and HBM saw 2.1x the bytes).  One pass: the image itself.
            // A GREY scene's three channels are equal: its planes hold one float per pixel (256 bytes per tile and pass), and
            // k_accumulate_tiled writes the sum to the three channels.
            if (GREY && P.pass_stride) {
              P.out[(size_t)pass * P.pass_stride + ((size_t)(ly >> 3) * tiles_x + (lx >> 3)) * 64u + (size_t)(((ly & 7u) << 3) + (lx & 7u))] = (float)rad0;
            } else {
              float *dst = P.pass_stride ? P.out + (size_t)pass * P.pass_stride + ((size_t)(ly >> 3) * tiles_x + (lx >> 3)) * 192u +
                                               (size_t)(((ly & 7u) << 3) + (lx & 7u)) * 3u
                                         : P.out + 3 * ((size_t)ly * (size_t)win_w + lx);
              dst[0] = (float)rad0;
              dst[1] = (float)rad1;
              dst[2] = (float)rad2;
            }
            if (P.tile_cost && pass == 0) // what this path cost, for the next launch's hand-out order
              atomicAdd(P.tile_cost + ((ly >> 3) * tiles_x + (lx >> 3)), n_nodes + n_tris + 16u * n_rays - cost_base);
          }
        }
        have_ray = false;
        want_pixel = path_done;
      }

#ifdef MGPU_UTIL
      cyc_sub[2] += clock64() - cyc_s; cyc_s = clock64();
#endif
      // ---- (2) path hand-out, executed by the whole wave (cursor variables are wave-uniform) ----
      // deferred start: with few lanes asking for a new path while others still traverse, the lanes stay parked (state
      // SHADE, no ray) and the path-start body runs later for more of them at once
      const bool defer = !exhausted && (cN + cT0) > 0 && __popcll(MGPU_BALLOT(want_pixel)) < (PRIM ? MGPU_START_MIN_PRIM : (LDS_SCENE ? MGPU_START_MIN_LDS : MGPU_START_MIN_HBM));
      for (;;) {
        const unsigned long long want = MGPU_BALLOT(want_pixel);
        if (!want || exhausted || defer) break;
        if (in_item >= 64) { // current item used up: take the next one from the workgroup's cursor
          uint32_t cur_shard = 0, item_local = 0;
#ifdef MGPU_EMU_STATS
          if (lane == 0) atomicAdd(&emu_stats[58], 1ull); // items taken (attempts included)
#endif
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
            // the workgroup's reservation is used up: one wave refills it, the others come back and retry
            uint32_t flag = 0;
            if (lane == 0) flag = __hip_atomic_load(&wg_dry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__builtin_amdgcn_readfirstlane((int)flag)) {
              exhausted = true;
#ifdef MGPU_UTIL
              cyc_dry = wall_clock64();
              dry_mark = true;
#endif
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
              if ((uint32_t)now >= (uint32_t)(now >> 32)) { // still empty (nobody refilled it while we took the lock)
                bool got = false;
                uint32_t off = wg_shard_off;
                while (off < (uint32_t)kShards) {
                  // a workgroup whose home part is used up moves on to the next part for good
                  const uint32_t sh = (home_shard + off) % (uint32_t)kShards;
                  const uint32_t base = atomicAdd(P.work_counter + sh, kWgChunk);
                  // index range of part sh: [0, n_sh)
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
          // tile and pass of the new item (tile-major: a tile's passes are consecutive items); with a cost order the
          // i-th tile handed out is tile_order[i]: most expensive first, so that the launch ends on cheap paths
          const uint32_t item = LDS_SCENE ? item_local * (uint32_t)kShards + cur_shard : cur_shard * shard_items + item_local;
          const uint32_t ti = item / (uint32_t)P.passes;
          item_pass = item - ti * (uint32_t)P.passes;
          item_tile = P.tile_order ? (uint32_t)__builtin_amdgcn_readfirstlane((int)P.tile_order[ti]) : ti;
          if constexpr (PRIM) { // the new item's primary rays, one per lane (every slot of the previous item has been copied out)
            const uint32_t tx = item_tile % tiles_x, ty = item_tile / tiles_x;
            const uint32_t x = tx * 8 + ((uint32_t)lane & 7u), y = ty * 8 + ((uint32_t)lane >> 3);
            if (x < (uint32_t)win_w && y < (uint32_t)P.n_rows) {
              const int gy = (P.y_first + (int)(y / (uint32_t)P.strip_h) * P.y_period + (int)(y % (uint32_t)P.strip_h)) * P.pix_step;
              const int gx = (P.x0 + (int)x) * P.pix_step;
              const uint32_t gpix = (uint32_t)gy * (uint32_t)P.W + (uint32_t)gx;
              uint32_t s4[4];
              if (P.rng_mode == MGPU_RNG_TABLE) {
                const uint4 q = reinterpret_cast<const uint4 *>(P.rng_states)[(size_t)item_pass * P.W * P.H + gpix];
                s4[0] = q.x; s4[1] = q.y; s4[2] = q.z; s4[3] = q.w;
              } else {
                hash_state(P.seed, P.pass_base + item_pass, gpix, s4);
              }
              Rng r{s4[0], s4[1], s4[2], s4[3]};
              const float ju = (float)(rng_next(r) - 0.5);
              const float jv = (float)(rng_next(r) - 0.5);
              const V3 d = camera_dir(P.frame, (double)((float)gx + ju), (double)((float)gy + jv));
              PrimRay &pr = s_prim[lane];
              pr.d[0] = d.x; pr.d[1] = d.y; pr.d[2] = d.z;
              pr.s[0] = r.x; pr.s[1] = r.y; pr.s[2] = r.z; pr.s[3] = r.w;
            }
            if constexpr (LDS_SCENE) {
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else { // device memory: the stores must have left the wave before other lanes load the slots (workgroup scope: one CU, one L1)
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
          }
        }
        if (want_pixel) {
          const uint32_t rank = __popcll(want & ((1ull << lane) - 1ull));
          const uint32_t slot = in_item + rank;
          if (slot < 64) {
            const uint32_t tx = item_tile % tiles_x, ty = item_tile / tiles_x;
            const uint32_t x = tx * 8 + (slot & 7), y = ty * 8 + (slot >> 3);
            if (x < (uint32_t)win_w && y < (uint32_t)P.n_rows) { // slots of an edge tile outside the window are skipped
              lx = x; ly = y;
              pass = (int)item_pass;
              have_path = true;
              want_pixel = false;
              if constexpr (PRIM) { // this lane's path has ended: its ray registers are free
                const PrimRay &pr = s_prim[slot];
                dir = v3(pr.d[0], pr.d[1], pr.d[2]);
                rng = Rng{pr.s[0], pr.s[1], pr.s[2], pr.s[3]};
              }
            }
          }
        }
        in_item += (uint32_t)__popcll(want);
        if (in_item >= 64) in_item = 64;
      }

      // ---- (3) next path / next traversal ----
      if (shade_lane) {
        if (path_done && have_path) {
          have_path = false;
#ifdef MGPU_EMU_STATS
          atomicAdd(&emu_stats[57], 1ull); // paths started
#endif
#ifdef MGPU_UTIL
          if (lane == __ffsll((long long)MGPU_BALLOT(1)) - 1) u_start_steps++;
#endif
          // start a new eye path (PathTrace prologue, render.cc:387-400)
          if constexpr (PRIM) {
            if (P.probe) {
              const int gy = (P.y_first + (int)(ly / (uint32_t)P.strip_h) * P.y_period + (int)(ly % (uint32_t)P.strip_h)) * P.pix_step;
              const uint32_t gpix = (uint32_t)gy * (uint32_t)P.W + (uint32_t)((P.x0 + (int)lx) * P.pix_step);
              probe_on = gpix == P.probe_pixel && (uint32_t)pass == P.probe_pass;
            }
            org = v3(P.frame[0], P.frame[1], P.frame[2]);
          } else {
          const uint32_t j = ly;
          // pix_step > 1: Render(step): the window is in units of step x step blocks and a block's path is that of its
          // top-left pixel (render.cc:657-681)
          const int gy = (P.y_first + (int)(j / (uint32_t)P.strip_h) * P.y_period + (int)(j % (uint32_t)P.strip_h)) * P.pix_step;
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
          probe_on = P.probe && gpix == P.probe_pixel && (uint32_t)pass == P.probe_pass;
          const float ju = (float)(rng_next(rng) - 0.5);
          const float jv = (float)(rng_next(rng) - 0.5);
          org = v3(P.frame[0], P.frame[1], P.frame[2]);
          dir = camera_dir(P.frame, (double)((float)gx + ju), (double)((float)gy + jv));
          }
          thr0 = 1.0;
          if (!GREY) thr1 = thr2 = 1.0;
          pathLength = 1;
          ++paths;
          cost_base = n_nodes + n_tris + 16u * n_rays;
          path_done = false;
        }
        if (path_done) {
          if (exhausted) st = ST_IDLE; // the work counter is exhausted: this lane is finished (else: parked, start deferred)
        } else {
#ifdef MGPU_UTIL
          cyc_sub[4] += clock64() - cyc_s; cyc_s = clock64();
#endif
          // arm the traversal of (org, dir): BVHAccel::Traverse prologue, bvh_accel.cc:774-802
          sgn = (dir.x < 0.0 ? 1u : 0u) | (dir.y < 0.0 ? 2u : 0u) | (dir.z < 0.0 ? 4u : 0u);
          const bool inv_ok = inverse_dir_w(dir, ix, iy, iz); // 1.0 / dir, no zero guard, as the reference
          ray_plain = sc.boxes_ordered && inv_ok && origin_is_finite(org);
          if (LDS_SCENE && MGPU_LEAF_HINTS) {
            // the rays leaf_hint_make's cones are sized for: an origin within the launch's reach of the scene's centre (camera rays by
            // construction, bounces that start inside the scene's box); their directions are no longer than 1 + 2^-10 in a scene
            // that gets hints at all (shading normals of length <= 1 + 2^-11, checked when the scene is created)
#ifdef MGPU_EXP_HINT_NOCHECK // (experiment, unsound: what the origin test costs)
            ray_hints = ray_plain && fabs(ix) < 0x1p100 && fabs(iy) < 0x1p100 && fabs(iz) < 0x1p100;
#else
            const double ox = org.x - P.hint_c[0], oy = org.y - P.hint_c[1], oz = org.z - P.hint_c[2];
            ray_hints = ray_plain && ox * ox + oy * oy + oz * oz <= P.hint_q2 && fabs(ix) < 0x1p100 && fabs(iy) < 0x1p100 && fabs(iz) < 0x1p100;
#endif
          }
          bt = kDblMax; bu = 0.0; bv = 0.0; bslot = kNoHit;
          sp = 0;
          have_ray = true;
          ++n_rays;
          st = ST_NODE;
          if constexpr (LDS_SCENE) {
#if MGPU_ROOT_AT_ARM
            // The reference's first pop -- the root, tested against the fresh ray (bvh_accel.cc:805-812) -- happens here, where
            // the ray is made: three rays in five of the Cornell frame (sky and plane pixels, bounces that leave the box) end at
            // this test, and no longer pass through a NODE step to find that out.  Same test, same count, same pushes.
            {
              const double2 rb0 = *reinterpret_cast<const double2 *>(lds_nodes), rb1 = *reinterpret_cast<const double2 *>(lds_nodes + 16),
                            rb2 = *reinterpret_cast<const double2 *>(lds_nodes + 32);
              const int4 rmeta = *reinterpret_cast<const int4 *>(lds_nodes + 48);
              const bool sx = (sgn & 1u) != 0u, sy = (sgn & 2u) != 0u, sz = (sgn & 4u) != 0u;
              ++n_nodes;
              sp = -1;
              st = ST_SHADE; // a miss: the ray is finished
              const bool rhit = (MGPU_BALLOT(!ray_plain) == 0ull) ? slab_hit<true>(rb0, rb1, rb2, org, ix, iy, iz, sx, sy, sz, bt)
                                                              : slab_hit<false>(rb0, rb1, rb2, org, ix, iy, iz, sx, sy, sz, bt);
              if (rhit) {
                if (rmeta.x == 0) {
                  const bool nearIsSecond = ((sgn >> (uint32_t)rmeta.y) & 1u) != 0u;
                  const uint32_t c0 = (uint32_t)rmeta.z, c1 = (uint32_t)rmeta.w;
                  stk.put(0, nearIsSecond ? c0 : c1); // far
                  stk.put(1, nearIsSecond ? c1 : c0); // near: popped first
                  sp = 1;
                  st = ST_NODE;
                } else if (rmeta.z != 0) {
                  tri_cur = (uint32_t)rmeta.w;
                  tri_end = (uint32_t)rmeta.w + (uint32_t)rmeta.z;
                  st = ST_TRI;
                }
              }
            }
#else
            stk.put(0, 0u);
#endif
          } else {
            cur = TL ? kWTreelet : sc.wroot; // the super root (record 0 of the treelet): its child 0 is the tree's root (the reference's first pop)
            n_nodes -= 1u;  // ... and its child 1 a dummy the reference never pops
          }
        }
      }
#ifdef MGPU_UTIL
      cyc_sub[5] += clock64() - cyc_s;
      if (dry_mark) { // lane-level snapshot right after the hand-out that found the cursor dry
        dry_mark = false;
        dry_rays = n_rays;
        dry_active = (st != ST_IDLE) ? 1u : 0u;
        dry_plen = (st != ST_IDLE) ? (uint32_t)pathLength : 0u;
      }
      if (cyc_dry) ++steps_s;
#endif
      MGPU_TOCK(cyc_shade);
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
#if MGPU_OCC
  __syncthreads();
  if (threadIdx.x < 8 && P.stats && s_occ[threadIdx.x]) atomicAdd(&P.stats[kOccNodeTrips + threadIdx.x], (unsigned long long)s_occ[threadIdx.x]);
#endif
#ifdef MGPU_UTIL
#ifdef MGPU_UTIL_HINTCLASS
  for (int k = 0; k < 4; ++k)
    if (u_hc[k]) atomicAdd(&P.stats[28 + k], u_hc[k]); // consultations | dropped tests << 32
#else
  if (u_hist)
    for (int k = 0; k < 4; ++k)
      if ((u_hist >> (16 * k)) & 0xffffull) atomicAdd(&P.stats[28 + k], (u_hist >> (16 * k)) & 0xffffull);
#endif
  {
    unsigned long long a = u_node, b = u_tri, cc = u_shade, d = u_shade_lanes;
    unsigned long long e_rays = n_rays - dry_rays, e_act = dry_active, e_plen = dry_plen, it_n = u_node_it, it_t = u_tri_it;
    unsigned long long s_tail = u_tail_steps, s_bounce = u_bounce_steps, s_start = u_start_steps, c_tail = cyc_sub[1], c_bounce = cyc_sub[3];
    for (int off = 32; off; off >>= 1) {
      a += __shfl_down(a, off);
      b += __shfl_down(b, off);
      cc += __shfl_down(cc, off);
      d += __shfl_down(d, off);
      e_rays += __shfl_down(e_rays, off);
      e_act += __shfl_down(e_act, off);
      e_plen += __shfl_down(e_plen, off);
      it_n += __shfl_down(it_n, off);
      it_t += __shfl_down(it_t, off);
      s_tail += __shfl_down(s_tail, off);
      s_bounce += __shfl_down(s_bounce, off);
      s_start += __shfl_down(s_start, off);
      c_tail += __shfl_down(c_tail, off);
      c_bounce += __shfl_down(c_bounce, off);
    }
    {
      unsigned long long h0 = u_hint_fresh, h1 = u_hint_dropped, h2 = u_hint_empty, h3 = u_hint_steps;
      for (int off = 32; off; off >>= 1) {
        h0 += __shfl_down(h0, off);
        h1 += __shfl_down(h1, off);
        h2 += __shfl_down(h2, off);
        h3 += __shfl_down(h3, off);
      }
      if (lane == 0) {
        atomicAdd(&P.stats[6], h0 | (h3 << 32));
        atomicAdd(&P.stats[7], h1 | (h2 << 32));
      }
    }
    if (lane == 0) {
      atomicAdd(&P.stats[kUtilNodeSteps], a);
      atomicAdd(&P.stats[kUtilTriSteps], b);
      atomicAdd(&P.stats[kUtilOuter], cc);
      atomicAdd(&P.stats[kUtilShadeLanes], d);
      atomicAdd(&P.stats[kUtilNodeLanes], it_n); // wave-level iterations of the NODE / TRI inner loops
      atomicAdd(&P.stats[kUtilTriLanes], it_t);
      atomicAdd(&P.stats[16], cyc_node);
      atomicAdd(&P.stats[17], cyc_tri);
      atomicAdd(&P.stats[18], cyc_shade);
      cyc_sub[1] = c_tail;   // wave totals (booked by the sub-body's first lane), not lane-0 samples
      cyc_sub[3] = c_bounce;
      for (int k = 0; k < 6; ++k) atomicAdd(&P.stats[19 + k], cyc_sub[k]);
      atomicAdd(&P.stats[kUtilTraceLanes], s_tail);
      atomicAdd(&P.stats[kUtilGenLanes], s_bounce);
      atomicAdd(&P.stats[5], s_start);
      const unsigned long long loop_cyc = clock64() - cyc_loop0;
      atomicAdd(&P.stats[25], loop_cyc);
      atomicMax(&P.stats[26], loop_cyc);
      atomicAdd(&P.stats[27], 1ull);
      if (P.wave_log) {
        const size_t wid = (size_t)blockIdx.x * kWaves + wave;
        if (wid < 16384) {
          unsigned xcc;
          MGPU_XCC_ID(xcc);
          P.wave_log[4 * wid + 0] = wall_loop0;
          P.wave_log[4 * wid + 1] = wall_clock64();
          P.wave_log[4 * wid + 2] = cyc_dry;
          P.wave_log[4 * wid + 3] = (xcc & 0xf) | (v1 << 8);
          // after-dry record, second half of the log: rays traced, lanes alive and their pathLength sum at dry, steps by kind
          unsigned long long *w2 = P.wave_log + 4 * 16384 + 4 * wid;
          w2[0] = e_rays;
          w2[1] = e_act | (e_plen << 16);
          w2[2] = (unsigned long long)steps_n | ((unsigned long long)steps_t << 20) | ((unsigned long long)steps_s << 40);
          w2[3] = dry_steps;
        }
      }
    }
  }
#endif
}

// =====================================================================================================================
// Pass accumulation: image[px] = pass 0 + pass 1 + ... in float32, in pass order (AccumImage, main_sdl.cc:138-143);
// count += passes.  `resume`: the planes hold a later group of passes and the sum continues from the image's current value.
// =====================================================================================================================
// One pass writes the image itself; only the counters are due (count[px]++, render.cc:677-679).
__global__ __launch_bounds__(256) void k_count_add(int32_t *__restrict__ count, size_t npix, int passes) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < npix) count[i] += passes;
}

void launch_count_add(hipStream_t s, int32_t *count, size_t npix, int passes) {
  hipLaunchKernelGGL(k_count_add, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, count, npix, passes);
}

// Planes in tile-major order (what k_render_sm writes for several passes, see there): slot ((ty * tiles_x + tx) * 64 +
// (y % 8) * 8 + x % 8) * 3 + channel of every plane holds image float 3 * (y * win_w + x) + channel of the window.  The
// threads walk the PLANES (consecutive threads read consecutive 16 bytes of all passes -- 16/17 of the traffic is these
// reads) and scatter 96-byte tile rows into the image.  VEC = 4: window widths that are multiples of 8 (a tile row then
// is 24 whole floats of one image row, 16-byte aligned on both sides); VEC = 1 otherwise.  Same additions per float, same
// order.
template <int VEC>
__global__ __launch_bounds__(256) void k_accumulate_tiled(const float *__restrict__ planes, size_t plane_stride, int passes,
                                                           uint32_t win_w, uint32_t n_rows, uint32_t tiles_x,
                                                           float *__restrict__ image, int32_t *__restrict__ count, bool resume) {
  const size_t p0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC; // float index inside a plane
  if (p0 >= plane_stride) return;
  const uint32_t tile = (uint32_t)(p0 / 192u), o = (uint32_t)(p0 - (size_t)tile * 192u);
  const uint32_t r = o / 24u, c = o - 24u * r; // row inside the tile, float inside that row (pixel c / 3, channel c % 3)
  const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint32_t y = ty * 8u + r, x = tx * 8u + c / 3u;
  if (y >= n_rows || x >= win_w) return; // padding of an edge tile
  const size_t f = ((size_t)y * win_w + (size_t)tx * 8u) * 3u + c;
  if (VEC == 4) {
    float4 acc = resume ? *reinterpret_cast<const float4 *>(image + f) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < passes; ++p) {
      const float4 v = *reinterpret_cast<const float4 *>(planes + (size_t)p * plane_stride + p0);
      acc.x += v.x;
      acc.y += v.y;
      acc.z += v.z;
      acc.w += v.w;
    }
    *reinterpret_cast<float4 *>(image + f) = acc;
    if (count) { // the pixels whose first channel lies in this thread's four floats
      for (uint32_t k = 0; k < 4; ++k)
        if ((c + k) % 3u == 0) count[(size_t)y * win_w + (size_t)tx * 8u + (c + k) / 3u] += passes;
    }
  } else {
    float acc = resume ? image[f] : 0.f;
    for (int p = 0; p < passes; ++p) acc += planes[(size_t)p * plane_stride + p0];
    image[f] = acc;
    if (count && c % 3u == 0) count[(size_t)y * win_w + x] += passes;
  }
}

// The same for planes of ONE float per pixel (grey scenes: the three channels of a pixel are equal in every pass, so their sums
// are): 64 floats per tile, the sum written to the pixel's three channels.  VEC = 4: four pixels of a tile row = one float4 per
// pass in, three float4 out.
template <int VEC>
__global__ __launch_bounds__(256) void k_accumulate_tiled_mono(const float *__restrict__ planes, size_t plane_stride, int passes,
                                                                uint32_t win_w, uint32_t n_rows, uint32_t tiles_x,
                                                                float *__restrict__ image, int32_t *__restrict__ count, bool resume) {
  const size_t p0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC; // float = pixel index inside a plane
  if (p0 >= plane_stride) return;
  const uint32_t tile = (uint32_t)(p0 >> 6), o = (uint32_t)(p0 & 63u);
  const uint32_t r = o >> 3, c = o & 7u;
  const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint32_t y = ty * 8u + r, x = tx * 8u + c;
  if (y >= n_rows || x >= win_w) return; // padding of an edge tile
  const size_t px = (size_t)y * win_w + x;
  if (VEC == 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (resume) {
      const float4 i0 = *reinterpret_cast<const float4 *>(image + 3 * px), i1 = *reinterpret_cast<const float4 *>(image + 3 * px + 4),
                   i2 = *reinterpret_cast<const float4 *>(image + 3 * px + 8);
      acc = make_float4(i0.x, i0.w, i1.z, i2.y);
    }
    for (int p = 0; p < passes; ++p) {
      const float4 v = *reinterpret_cast<const float4 *>(planes + (size_t)p * plane_stride + p0);
      acc.x += v.x;
      acc.y += v.y;
      acc.z += v.z;
      acc.w += v.w;
    }
    *reinterpret_cast<float4 *>(image + 3 * px) = make_float4(acc.x, acc.x, acc.x, acc.y);
    *reinterpret_cast<float4 *>(image + 3 * px + 4) = make_float4(acc.y, acc.y, acc.z, acc.z);
    *reinterpret_cast<float4 *>(image + 3 * px + 8) = make_float4(acc.z, acc.w, acc.w, acc.w);
    if (count)
      for (uint32_t k = 0; k < 4; ++k) count[px + k] += passes;
  } else {
    float acc = resume ? image[3 * px] : 0.f;
    for (int p = 0; p < passes; ++p) acc += planes[(size_t)p * plane_stride + p0];
    image[3 * px] = image[3 * px + 1] = image[3 * px + 2] = acc;
    if (count) count[px] += passes;
  }
}

void launch_accumulate_tiled(hipStream_t s, const float *planes, size_t plane_stride, bool mono, int passes, size_t n_floats, int win_w,
                             float *image, int32_t *count, bool resume) {
  const uint32_t tiles_x = (uint32_t)(win_w + 7) >> 3;
  const uint32_t n_rows = (uint32_t)(n_floats / ((size_t)3 * win_w));
  if (mono) {
    if (win_w % 8 == 0 && ((uintptr_t)image & 15) == 0 && ((uintptr_t)planes & 15) == 0 && plane_stride % 4 == 0) {
      hipLaunchKernelGGL(k_accumulate_tiled_mono<4>, dim3((unsigned)((plane_stride / 4 + 255) / 256)), dim3(256), 0, s, planes, plane_stride,
                         passes, (uint32_t)win_w, n_rows, tiles_x, image, count, resume);
    } else {
      hipLaunchKernelGGL(k_accumulate_tiled_mono<1>, dim3((unsigned)((plane_stride + 255) / 256)), dim3(256), 0, s, planes, plane_stride,
                         passes, (uint32_t)win_w, n_rows, tiles_x, image, count, resume);
    }
    return;
  }
  if (win_w % 8 == 0 && ((uintptr_t)image & 15) == 0 && ((uintptr_t)planes & 15) == 0 && plane_stride % 4 == 0) {
    hipLaunchKernelGGL(k_accumulate_tiled<4>, dim3((unsigned)((plane_stride / 4 + 255) / 256)), dim3(256), 0, s, planes, plane_stride,
                       passes, (uint32_t)win_w, n_rows, tiles_x, image, count, resume);
  } else {
    hipLaunchKernelGGL(k_accumulate_tiled<1>, dim3((unsigned)((plane_stride + 255) / 256)), dim3(256), 0, s, planes, plane_stride,
                       passes, (uint32_t)win_w, n_rows, tiles_x, image, count, resume);
  }
}

// =====================================================================================================================
// k_order_tiles: hand-out order of the next launch = tiles by descending cost in the previous one.
// A persistent launch ends when its last path ends, and a path that bounces four times is in flight ~10x longer than
// one that leaves through the sky; with tiles handed out in image order the launch drains for up to ~1 ms on whatever
// expensive paths happened to come last.  Longest-processing-time-first: expensive tiles go first, and the waves chew
// on sky tiles while their last expensive paths finish.
// Counting sort on 256 logarithmic cost buckets, one 1024-thread workgroup; the order inside a bucket is arbitrary,
// which is harmless: the image does not depend on the order (per-(pixel, pass) RNG).  All-zero costs (first launch of
// a layout) give the image order.  cost[] is zeroed for the next launch.
// =====================================================================================================================
__global__ __launch_bounds__(1024) void k_order_tiles(uint32_t *__restrict__ cost, uint32_t n, uint32_t *__restrict__ order) {
  __shared__ uint32_t hist[256], start[256];
  __shared__ uint32_t any;
  const uint32_t tid = threadIdx.x;
  if (tid < 256) hist[tid] = 0;
  if (tid == 0) any = 0;
  __syncthreads();
  // bucket = 8 * floor(log2(cost)) + next three bits, clamped to 255; larger cost -> larger bucket
  auto bucket = [](uint32_t c) -> uint32_t {
    if (c < 8) return c;
    const uint32_t e = 31u - (uint32_t)__clz((int)c);
    const uint32_t b = 8u * (e - 2u) + ((c >> (e - 3u)) & 7u);
    return b > 255u ? 255u : b;
  };
  // both passes read cost[] eight elements per thread at a time, so that a batch's loads are in flight together (one
  // workgroup has nobody to hide a load's latency behind).  35 us for the 32 400 tiles of a 1080p frame, most of it the
  // LDS atomics on the few buckets a typical image fills.
  uint32_t seen = 0;
  for (uint32_t i0 = tid; i0 < n; i0 += 8 * 1024) {
    uint32_t c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = (i0 + k * 1024 < n) ? cost[i0 + k * 1024] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k * 1024 < n) {
        seen |= c[k];
        atomicAdd(&hist[bucket(c[k])], 1u);
      }
  }
  if (seen) any = 1;
  __syncthreads();
  if (!any) {
    for (uint32_t i = tid; i < n; i += 1024) order[i] = i;
    return;
  }
  if (tid < 64) { // exclusive scan over the buckets, most expensive first: lane l owns buckets 255-4l .. 252-4l
    const uint32_t b0 = 255u - 4u * tid;
    const uint32_t h0 = hist[b0], h1 = hist[b0 - 1], h2 = hist[b0 - 2], h3 = hist[b0 - 3];
    uint32_t incl = h0 + h1 + h2 + h3;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(incl, off);
      if ((int)tid >= off) incl += v;
    }
    const uint32_t excl = incl - (h0 + h1 + h2 + h3);
    start[b0] = excl;
    start[b0 - 1] = excl + h0;
    start[b0 - 2] = excl + h0 + h1;
    start[b0 - 3] = excl + h0 + h1 + h2;
  }
  __syncthreads();
  for (uint32_t i0 = tid; i0 < n; i0 += 8 * 1024) {
    uint32_t c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = (i0 + k * 1024 < n) ? cost[i0 + k * 1024] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k * 1024 < n) {
        order[atomicAdd(&start[bucket(c[k])], 1u)] = i0 + k * 1024;
        cost[i0 + k * 1024] = 0;
      }
  }
}

// k_order_tiles_z: the same hand-out order for scenes whose BVH stays in HBM, where WHICH tiles an XCD's waves work on together
// decides what its 4 MB of L2 holds: tiles along the Z-order (Morton) curve of the tile grid, stably split into cost classes --
// class = octaves below the most expensive tile, at most kZClasses -- expensive classes first.  Inside a class neighbours on the
// curve stay neighbours in the hand-out, so a contiguous eighth of the order (what one XCD draws from, see the work cursor) is a
// few compact image regions instead of whatever the atomics of the 256-bucket sort left next to each other.  classes == 1: the
// pure curve.  One 1024-thread workgroup; thread t owns codes [t * per, (t + 1) * per) of the curve in both passes, which makes
// the placement deterministic.  cost[] is zeroed for the next launch.
constexpr int kZClasses = 8;
__device__ __forceinline__ uint32_t z_compact(uint32_t v) { // every second bit of v, packed
  v &= 0x55555555u;
  v = (v | (v >> 1)) & 0x33333333u;
  v = (v | (v >> 2)) & 0x0F0F0F0Fu;
  v = (v | (v >> 4)) & 0x00FF00FFu;
  v = (v | (v >> 8)) & 0x0000FFFFu;
  return v;
}
__global__ __launch_bounds__(1024) void k_order_tiles_z(uint32_t *__restrict__ cost, uint32_t tiles_x, uint32_t tiles_y, int classes,
                                                        uint32_t *__restrict__ order) {
  __shared__ uint32_t cnt[kZClasses][1024];
  __shared__ uint32_t class_base[kZClasses];
  __shared__ uint32_t max_cost;
  const uint32_t tid = threadIdx.x, n = tiles_x * tiles_y;
  uint32_t side = 1;
  while (side < tiles_x || side < tiles_y) side <<= 1;
  const uint32_t codes = side * side, per = (codes + 1023u) / 1024u;
  if (tid == 0) max_cost = 0;
  __syncthreads();
  if (classes > 1 || classes < 0) {
    uint32_t m = 0;
    for (uint32_t i = tid; i < n; i += 1024) m = max(m, cost[i]);
    for (int off = 32; off; off >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, off));
    if ((tid & 63u) == 0u) atomicMax(&max_cost, m);
    __syncthreads();
  }
  const uint32_t top = max_cost ? 31u - (uint32_t)__clz((int)max_cost) : 0u; // floor(log2) of the largest cost
  // classes < 0: two classes, the cheapest -classes per cent of the tiles (by the 256-bucket histogram of k_order_tiles) last
  __shared__ uint32_t hist[256], tail_bucket;
  auto bucket = [](uint32_t c) -> uint32_t {
    if (c < 8) return c;
    const uint32_t e = 31u - (uint32_t)__clz((int)c);
    const uint32_t b = 8u * (e - 2u) + ((c >> (e - 3u)) & 7u);
    return b > 255u ? 255u : b;
  };
  if (classes < 0) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 1024) atomicAdd(&hist[bucket(cost[i])], 1u);
    __syncthreads();
    if (tid == 0) {
      const uint32_t want = (uint32_t)(((uint64_t)n * (uint64_t)(-classes)) / 100u);
      uint32_t run = 0, b = 0;
      while (b < 256u && run + hist[b] <= want) run += hist[b++];
      tail_bucket = b; // buckets below b form the tail (none when the cheapest bucket alone is larger than the share asked for)
    }
    __syncthreads();
  }
  auto cls = [&](uint32_t c) -> uint32_t {
    if (classes < 0) return (max_cost != 0u && bucket(c) < tail_bucket) ? 1u : 0u;
    if (classes <= 1 || max_cost == 0u) return 0u;
    const uint32_t e = c ? 31u - (uint32_t)__clz((int)c) : 0u;
    return min(top - e, (uint32_t)classes - 1u);
  };
  uint32_t mine[kZClasses];
  for (int k = 0; k < kZClasses; ++k) mine[k] = 0;
  for (uint32_t code = tid * per; code < min((tid + 1u) * per, codes); ++code) {
    const uint32_t tx = z_compact(code), ty = z_compact(code >> 1);
    if (tx < tiles_x && ty < tiles_y) {
      const uint32_t k = cls(cost[ty * tiles_x + tx]);
      for (int q = 0; q < kZClasses; ++q) mine[q] += (q == (int)k) ? 1u : 0u; // (no dynamic indexing of a register array)
    }
  }
  for (int k = 0; k < kZClasses; ++k) cnt[k][tid] = mine[k];
  __syncthreads();
  // exclusive scan over the threads, class by class: wave k takes class k (1024 counts = 16 per lane)
  if ((tid >> 6) < (uint32_t)kZClasses) {
    const uint32_t k = tid >> 6, lane = tid & 63u;
    uint32_t v[16], sum = 0;
    for (int j = 0; j < 16; ++j) {
      v[j] = cnt[k][lane * 16 + j];
      sum += v[j];
    }
    uint32_t incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
      if ((int)lane >= off) incl += o;
    }
    uint32_t run = incl - sum;
    for (int j = 0; j < 16; ++j) {
      cnt[k][lane * 16 + j] = run;
      run += v[j];
    }
    if (lane == 63u) class_base[k] = incl; // the class's total, for now
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0;
    for (int k = 0; k < kZClasses; ++k) {
      const uint32_t t = class_base[k];
      class_base[k] = run;
      run += t;
    }
  }
  __syncthreads();
  uint32_t at[kZClasses];
  for (int k = 0; k < kZClasses; ++k) at[k] = class_base[k] + cnt[k][tid];
  for (uint32_t code = tid * per; code < min((tid + 1u) * per, codes); ++code) {
    const uint32_t tx = z_compact(code), ty = z_compact(code >> 1);
    if (tx < tiles_x && ty < tiles_y) {
      const uint32_t tile = ty * tiles_x + tx, k = cls(cost[tile]);
      uint32_t pos = 0;
      for (int q = 0; q < kZClasses; ++q)
        if (q == (int)k) pos = at[q]++;
      order[pos] = tile;
    }
  }
  __syncthreads(); // every cost has been read twice by now
  for (uint32_t i = tid; i < n; i += 1024) cost[i] = 0;
}

void launch_order_tiles(hipStream_t s, uint32_t *cost, uint32_t n_tiles, uint32_t *order, uint32_t tiles_x, uint32_t tiles_y, int z_classes) {
  if (z_classes != 0 && (uint64_t)tiles_x * tiles_y == n_tiles && tiles_x <= 32768u && tiles_y <= 32768u)
    hipLaunchKernelGGL(k_order_tiles_z, dim3(1), dim3(1024), 0, s, cost, tiles_x, tiles_y, z_classes > kZClasses ? kZClasses : (z_classes < -99 ? -99 : z_classes), order);
  else
    hipLaunchKernelGGL(k_order_tiles, dim3(1), dim3(1024), 0, s, cost, n_tiles, order);
}

// =====================================================================================================================
// k_tonemap: the display transforms of the reference's two drivers, fused with the 1/count normalisation
//   mode 0: HDRToLDR + fclamp of the console driver (main_console.cc:25-43): RGB8, out = clamp(int((in / count) * 255.5))
//   mode 1: Display + fclamp of the SDL driver (main_sdl.cc:157-165,420-477): BGRA8, gamma 2.2,
//           out = clamp(int(powf((1.0f / count) * in, 1.0f / 2.2f) * 255.5))
// The float -> double -> int conversion follows x86's cvttsd2si (out-of-range and NaN give INT_MIN, i.e. 0 after clamp).
// =====================================================================================================================
__device__ __forceinline__ unsigned char to_byte(double scaled) {
  int i;
  if (!(scaled < 2147483648.0) || scaled < -2147483648.0) i = INT_MIN; // also catches NaN
  else i = (int)scaled;
  return (unsigned char)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

__global__ __launch_bounds__(256) void k_tonemap(const float *__restrict__ image, const int32_t *__restrict__ count,
                                                  size_t npix, int mode, unsigned char *__restrict__ out) {
  const size_t px = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (px >= npix) return;
  const int c = count[px];
  const float r = image[3 * px + 0], g = image[3 * px + 1], b = image[3 * px + 2];
  if (mode == 0) {
    out[3 * px + 0] = to_byte((double)(r / (float)c) * 255.5);
    out[3 * px + 1] = to_byte((double)(g / (float)c) * 255.5);
    out[3 * px + 2] = to_byte((double)(b / (float)c) * 255.5);
  } else {
    const float scale = 1.0f / (float)c;
    const float inv_gamma = 1.0f / 2.2f;
    out[4 * px + 2] = to_byte((double)powf(scale * r, inv_gamma) * 255.5);
    out[4 * px + 1] = to_byte((double)powf(scale * g, inv_gamma) * 255.5);
    out[4 * px + 0] = to_byte((double)powf(scale * b, inv_gamma) * 255.5);
    out[4 * px + 3] = 255;
  }
}

void launch_tonemap(hipStream_t s, const float *image, const int32_t *count, size_t npix, int mode, unsigned char *out) {
  hipLaunchKernelGGL(k_tonemap, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, image, count, npix, mode, out);
}

// =====================================================================================================================
// launcher
// =====================================================================================================================
template <typename SE, bool LDS, int BLOCK, bool GREY, bool PRIM>
static hipError_t launch_grey(dim3 grid, hipStream_t s, size_t shmem, const DScene &sc, const RenderParams &p) {
  auto kern = k_render_sm<SE, LDS, BLOCK, GREY, PRIM>;
  // per device: dynamic-LDS size already granted to this instantiation.  Scenes on different devices are driven from
  // different host threads (and the multi-GPU frame drives several from one): the table is guarded.
  static size_t granted[16] = {0};
  static std::mutex granted_mutex;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (shmem > 48 * 1024) {
    std::lock_guard<std::mutex> lock(granted_mutex);
    if (dev < 0 || dev >= 16 || shmem > granted[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)shmem);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) granted[dev] = shmem;
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(BLOCK), shmem, s, sc, p);
  return hipGetLastError();
}

template <typename SE, bool LDS, int BLOCK, bool PRIM = false>
static hipError_t launch_one(dim3 grid, hipStream_t s, size_t shmem, const DScene &sc, const RenderParams &p) {
  return sc.grey ? launch_grey<SE, LDS, BLOCK, true, PRIM>(grid, s, shmem, sc, p) : launch_grey<SE, LDS, BLOCK, false, PRIM>(grid, s, shmem, sc, p);
}

// LDS the LDS-resident variant wants behind the scene for a wave's 64 staged primary rays (40 bytes each), 16 waves
size_t render_sm_prim_bytes() { return MGPU_PRIM_LDS ? (size_t)16 * 64 * 40 : 0; }

// Instantiations: LDS-resident scene (one 1024-thread workgroup per CU; stack entries of 1 or 2 bytes, see lds_stack_entry_bytes;
// with or without the primary-ray staging); HBM-resident scene (wide form, its own stacks) with 256-thread workgroups or one
// 1024-thread workgroup + treelet.
hipError_t launch_render_sm(int stack_entry_bytes, bool lds_scene, bool prim, int block, dim3 grid, hipStream_t s, size_t shmem, const DScene &sc,
                            const RenderParams &p) {
  if (lds_scene && block == 1024) {
    if (stack_entry_bytes == 1) return prim ? launch_one<uint8_t, true, 1024, true>(grid, s, shmem, sc, p) : launch_one<uint8_t, true, 1024>(grid, s, shmem, sc, p);
    if (stack_entry_bytes == 2) return prim ? launch_one<uint16_t, true, 1024, true>(grid, s, shmem, sc, p) : launch_one<uint16_t, true, 1024>(grid, s, shmem, sc, p);
  }
  if (!lds_scene && block == 256) return launch_one<uint32_t, false, 256>(grid, s, shmem, sc, p); // wide form: one variant
  if (!lds_scene && block == 1024 && sc.treelet) // ... + treelet in LDS; prim: the items' primary rays staged in device memory (RenderParams::prim_stage)
    return (prim && sc.grey && p.prim_stage) ? launch_grey<uint32_t, false, 1024, true, true>(grid, s, shmem, sc, p) // (a scene of coloured materials keeps the plain variant: staged, it spills 23 registers)
                                             : launch_one<uint32_t, false, 1024>(grid, s, shmem, sc, p);
#ifdef MGPU_EXP_768
  if (!lds_scene && block == 768 && sc.treelet) return launch_one<uint32_t, false, 768>(grid, s, shmem, sc, p); // experiment: 3 waves per SIMD, 168 VGPRs
#endif
  return hipErrorInvalidConfiguration;
}

} // namespace mgpu
