// mgpu_kernels.hpp -- launch-side declarations shared by mgpu_kernels.hip and mgpu_api.hip
#pragma once

#include <stdint.h>

#include "../../include/mgpu.h"
#include "mgpu_device.hpp"

namespace mgpu {

constexpr int kBlock = 256;      // 4 waves per workgroup
constexpr int kChunkTiles = 2;   // k_render v1: 8x8-pixel tiles handed to a wave per global-counter fetch
constexpr int kShards = 8;       // k_render_sm: work counters per launch = XCDs of an MI355X
constexpr int kSampleEvery = 8;  // k_render_sm: lane-occupancy accounting on every 8th NODE / TRI / SHADE step

// device-side statistics words (unsigned long long each)
enum : int { kStatTraceCalls = 0, kStatRays = 1, kStatNodes = 2, kStatTris = 3, kStatPaths = 4,
              // wave-level utilisation probes, filled only by -DMGPU_UTIL builds (scratch experiments)
              kUtilNodeSteps = 8, kUtilNodeLanes = 9, kUtilTriSteps = 10, kUtilTriLanes = 11, kUtilOuter = 12,
              kUtilTraceLanes = 13, kUtilShadeLanes = 14, kUtilGenLanes = 15,
              // k_render_sm, always on: the active-lane fraction of its three bodies measured in the run itself.  One step in
              // kSampleEvery (picked by the shader clock's low bits) is booked: wave-level trips of the body's loop and the
              // lanes active summed over those trips (a few compares after the loop; SHADE: the step and the lanes in it),
              // and the number of steps booked per body.
              kOccNodeTrips = 32, kOccNodeLanes = 33, kOccTriTrips = 34, kOccTriLanes = 35, kOccShadeSteps = 36,
              kOccShadeLanes = 37, kOccNodeBooked = 38, kOccTriBooked = 39, kStatWords = 40 };

struct RenderParams {
  double frame[12]; // origin, corner, du, dv  (Camera::BuildCameraFrame, camera.cc:40-220)
  float plane[4];
  double plane_n[3]; // normalize((double)plane[0..2]), the normal Plane::intersect returns (prim-plane.cc:30-33); host-computed
  int has_plane;
  int W, H;         // full frame (RNG tables and hash seeds index the full frame)
  int x0, x1;       // window columns
  int y_first, strip_h, y_period, n_rows; // row strips: local row j -> y_first + (j/strip_h)*y_period + j%strip_h
  int maxPathLength, passes;
  int pix_step;     // 1, or Render()'s `step`: window coordinates then count step x step blocks (k_render_sm only)
  // RN(1 / L) for L = 0 .. 16 (host division; [0] unused): the post-miss tail's `x / L` as q = x * y, q += fma(-q, L, x) * y,
  // which is the correctly rounded quotient when y is the correctly rounded reciprocal (Markstein); tests/test_host_cpu.py
  // checks the sequence against exact rational arithmetic.  Longer paths divide.
  double inv_len[17];
  // tail_unit[m][L0], 1 <= L0 <= maxPathLength <= 16: the post-miss tail's sum for throughput 1 starting at length L0 with the
  // material multiplier 0.5 applied (m = 1) or not (m = 0: the plane) -- the kernel's own loop run on the host
  // (fill_tail_unit): for a power-of-two throughput the sum is throughput x this, bit for bit.
  double tail_unit[2][17];
  int rng_mode;
  const uint32_t *rng_states; // device, MGPU_RNG_TABLE layout, or null
  unsigned long long seed;
  uint32_t pass_base;
  float *image;     // device, 3 * n_rows * (x1-x0)  (k_render v1 writes the pass-ordered sum here itself)
  int32_t *count;   // device or null
  float *out;       // k_render_sm: where per-pass radiance goes: pass planes (passes > 1) or the image (passes == 1)
  size_t pass_stride; // floats between consecutive pass planes of `out` (0 when passes == 1)
  uint32_t *work_counter;        // device, kShards words (k_render v1 uses the first), zeroed before the launch
  unsigned long long *stats;     // device, kStatWords, accumulated
  // Cost-ordered hand-out (k_render_sm): tile_order[i] = the tile handed out i-th (most expensive first, from the
  // previous launch's costs), or null = image order; tile_cost[tile] accumulates this launch's cost of pass 0.
  const uint32_t *tile_order;
  uint32_t *tile_cost;
  uint32_t lds_nodes_bytes, lds_tris_bytes; // k_render_sm<LDS_SCENE>: bytes of nodes / triangles staged into LDS
  uint32_t stack_cap;            // k_render_sm<LDS_SCENE>: stack entries per lane in LDS = tree depth + 1
  uint32_t lds_hint_cap;         // k_render_sm<LDS_SCENE>: leaf hint records (kHintFloats floats) that fit behind the scene (0: none), see kHintMinTris
  // ... and which rays may consult them (mgpu_device.hpp, leaf_hint_make): origins with |org - hint_c|^2 <= hint_q2 = hint_q^2, which
  // bounds |org - p0| of such a ray against every triangle and sizes pads and cones (host: render_frames_impl)
  double hint_c[3], hint_q2, hint_q;
  // k_render_sm<!LDS_SCENE, PRIM>: where the waves stage the primary rays of their current work item when LDS has no room for them
  // (the HBM-resident scene: stacks + treelet fill it) -- 64 x 40 bytes per wave of the launch in device memory, written and read by
  // that wave alone (its lines stay in the CU's L1 / L2)
  unsigned char *prim_stage;
  unsigned long long *wave_log;  // device or null: 4 words per wave (diagnostic builds only)
  double *probe;                 // device or null: kProbeStride doubles per PathTrace iteration of ONE path
  uint32_t probe_pixel, probe_pass; // full-frame pixel index and pass of the probed path
};
// the GREY tail loop of k_render_sm (render.cc:409-418 after the first miss) for throughput 1, diffuse 0.5
inline void fill_tail_unit(RenderParams &P) {
  for (int m = 0; m < 2; ++m)
    for (int L0 = 0; L0 <= 16; ++L0) {
      double rad = 0.0, thr = 1.0;
      if (L0 >= 1 && L0 <= P.maxPathLength && P.maxPathLength <= 16)
        for (int L = L0;; ++L) {
          const double x = thr * 0.5, y = P.inv_len[L], dl = (double)(unsigned)L;
          const double q = x * y;
          rad += __builtin_fma(__builtin_fma(-q, dl, x), y, q);
          if (L >= P.maxPathLength) break;
          if (m) thr *= 0.5;
        }
      P.tail_unit[m][L0] = rad;
    }
}
constexpr int kProbeStride = 16; // org[3] dir[3] t hit slot normal[3] materialID pathLength throughput.x radiance.x

// stack capacities (LDS entries per lane) the kernels are instantiated for
// `select` (device word or null): when given, the kernel runs only if k_trace_probe wrote its own id there
constexpr uint32_t kTraceSelectV1 = 0, kTraceSelectSm = 1;
void launch_trace(int cap, dim3 grid, hipStream_t s, const DScene &sc, const MgpuRay *rays, size_t n,
                  MgpuIntersection *out, uint8_t *hit, unsigned long long *stats, const uint32_t *select);
// k_trace_probe: samples 128 groups of 64 consecutive rays and writes kTraceSelectV1 to *select when at least three
// quarters of them are coherent (every direction within ~14 degrees of the group's first), kTraceSelectSm otherwise
void launch_trace_probe(hipStream_t s, const MgpuRay *rays, size_t n, uint32_t *select);
void launch_render(int cap, dim3 grid, hipStream_t s, const DScene &sc, const RenderParams &p);
int pick_stack_cap(int needed_entries);
// wave-scheduled state-machine renderer (mgpu_render_sm.hip); shmem = stacks (+ scene when lds_scene)
hipError_t launch_render_sm(int stack_entry_bytes, bool lds_scene, bool prim, int block, dim3 grid, hipStream_t s, size_t shmem, const DScene &sc,
                            const RenderParams &p);
// k_render_w5 (mgpu_render_w5.hip): the HBM-resident walk with its state divided by hand for five waves per SIMD.  block = 640 | 320;
// shmem = block / 64 * render_w5_wave_bytes() + the treelet's bytes (p.lds_nodes_bytes)
hipError_t launch_render_w5(int block, dim3 grid, hipStream_t s, size_t shmem, const DScene &sc, const RenderParams &p);
size_t render_w5_wave_bytes();   // LDS per wave: far-child stack + SHADE-only state
int render_w5_stack_entries();   // far-child stack entries per lane it keeps in LDS
size_t render_sm_prim_bytes(); // LDS the LDS-resident variant wants behind the scene for its primary-ray staging (0: compiled out)
// LDS-resident scene: a stack entry is a node index -- 1 byte up to 256 nodes, 2 up to 65 536 (larger trees never fit) --
// and a lane needs tree depth + 1 of them (bvh_accel.cc:805-834: a pop, then at most two pushes per level)
inline int lds_stack_entry_bytes(size_t nn) { return nn <= 256 ? 1 : (nn <= 65536 ? 2 : 4); }
__host__ __device__ inline size_t lds_stack_bytes(size_t waves, size_t cap, size_t entry_bytes) {
  return (waves * cap * 64 * entry_bytes + 15) & ~(size_t)15;
}
// k_render_env (mgpu_render_env.hip): RenderPanoramic
struct EnvParams {
  double origin[3];        // Camera::origin_ (BuildCameraFrame)
  double cos_psi, sin_psi; // cos / sin of atan2(0.5, 4.0), the stereo toe-in (camera.cc:311), from the host's libm
  int W, H;                // full frame: the spherical angles and the RNG states index the full frame
  int x0, y0, win_w, win_h; // window; image / count are window-local
  int maxPathLength, samples, stereo;
  int rng_mode;
  const uint32_t *rng_states; // device, 4 words per pixel of the full frame, or null
  unsigned long long seed;
  uint32_t pass_base;
  float *image;   // device, 3 * win_w * win_h, overwritten
  int32_t *count; // device, win_w * win_h, += samples; may be null
  uint32_t *work_counter; // one zeroed word
  unsigned long long *stats;
  uint32_t lds_nodes_bytes, lds_tris_bytes; // LDS_SCENE variant: bytes of nodes / triangles staged into LDS
  // tail_sum[L0] = the radiance of a path whose first miss comes at length L0 (2 <= L0 <= maxPathLength <= 32): 0.5 / L
  // added for L = L0 .. maxPathLength in that order, evaluated on the host with the same IEEE additions and divisions the
  // kernel's loop would perform (render.cc:563-574) -- one LDS read instead of up to fifteen divisions.  Longer paths loop.
  double tail_sum[33];
};
hipError_t launch_render_env(int cap, bool lds_scene, dim3 grid, hipStream_t s, const DScene &sc, const EnvParams &p);
// k_render_aov (mgpu_kernels.hip): ShowNormal (mode 0) / ShowUV (mode 1), one primary ray per pixel of the whole frame
struct AovParams {
  double frame[12];
  int W, H, mode, rng_mode;
  const uint32_t *rng_states; // device, 4 words per pixel, or null
  unsigned long long seed;
  uint32_t pass_base;
  float *image;   // device, 3 * W * H, overwritten
  int32_t *count; // device or null, += 1
  unsigned long long *stats;
};
void launch_render_aov(int cap, dim3 grid, hipStream_t s, const DScene &sc, const AovParams &p);
// k_stream_states (mgpu_stream.hip): MGPU_RNG_STREAM -- the start state of every (pass, pixel) in the reference's own
// serial stream, written to an MGPU_RNG_TABLE table; `state` (device, 4 words) is the stream state, in and out
constexpr int kStreamJumpBits = 44;      // T^(2^j), j < 44: a pass of 2^32 pixels at maxPathLength 341 draws < 2^43 numbers
constexpr int kStreamSerialJumpBits = 18; // the one-workgroup kernel: window offsets < 2^18 <=> maxPathLength <= kStreamMaxPathLength
constexpr int kStreamMaxPathLength = 341; // 256 * (2 + 3 * 340) < 2^18
struct StreamParams {
  double frame[12];
  float plane[4];
  double plane_n[3];
  int has_plane;
  int W, H, maxPathLength, passes;
  const uint4 *jump; // device: kStreamJumpBits x 128 columns (stream_jump_matrices)
  uint32_t *state;   // device: 4 words, in / out
  uint32_t *table;   // device: passes * W * H * 4 words
};
hipError_t launch_stream_states(int cap, hipStream_t s, const DScene &sc, const StreamParams &p);
// the chip-wide resolution of the same table (mgpu_stream.hip): scratch kept with the scene, the classification of a camera's
// pixels cached in it from call to call
struct StreamScratch {
  unsigned char *cls = nullptr;        // [npix] 0 / 1 / 2
  uint32_t *C = nullptr, *J = nullptr; // [npix] certain hits / uncertain pixels before a pixel
  uint32_t *U = nullptr;               // [npix] the uncertain pixels, in order
  uint4 *base = nullptr;               // [npix]
  unsigned long long *block_sum = nullptr; // [npix / 1024 + 1]
  unsigned char *F = nullptr, *uflag = nullptr;
  uint32_t *Sarr = nullptr, *USx = nullptr, *totals = nullptr, *bad = nullptr;
  size_t npix_cap = 0, sarr_cap = 0;
  // kept from call to call (round 6; a 1080p pass's table is 33 MB: allocating and freeing it, the 90 KB of jump matrices and the state word
  // on every Render() call cost three hipMalloc / hipFree pairs, each hipFree a device synchronisation)
  uint32_t *table = nullptr;           // start states, 16 bytes per pixel and pass
  size_t table_bytes = 0;
  uint32_t *state = nullptr;           // the stream state the kernels advance
  uint4 *jump = nullptr;               // T^(2^j), uploaded once
  // what the cached classification belongs to
  double key_frame[12];
  float key_plane[4];
  int key_has_plane = -1, key_W = 0, key_H = 0;
};
size_t stream_scratch_sarr_cap(size_t npix);
size_t stream_scratch_f_bytes();
// *unsettled_out: the verification asked for more than 64 repetitions -- no table was written; the caller takes k_stream_states
hipError_t stream_states_resolve(int cap, hipStream_t st, const DScene &sc, const StreamParams &p, StreamScratch &scratch, int num_cu, bool fresh_camera,
                                 uint32_t *retries_out, bool *unsettled_out);
void stream_jump_matrices(uint32_t *out /* kStreamJumpBits * 128 * 4 words */);
// k_trace_sm (mgpu_trace_sm.hip): persistent, wave-scheduled batched trace; `counter` = one zeroed device word
hipError_t launch_trace_sm(dim3 grid, hipStream_t s, const DScene &sc, const MgpuRay *rays, uint32_t n,
                           MgpuIntersection *out, uint8_t *hit, uint32_t *counter, unsigned long long *stats,
                           const uint32_t *select);
// k_trace_server (mgpu_trace_server.hip): resident traversal for one-ray-per-call callers.  The mailbox lives in host memory
// the device maps (hipHostMallocMapped | hipHostMallocCoherent); every word has ONE writer.
constexpr int kSrvWaves = 16;         // one 64-thread workgroup each
constexpr int kSrvSlotsPerWave = 16;  // the 64 bytes of request numbers a wave polls with one load
constexpr int kSrvSlots = kSrvWaves * kSrvSlotsPerWave;
struct alignas(64) TraceMailbox {
  uint32_t req[kSrvSlots];  // host: number of the slot's latest request (a caller owns the slot while its call lasts)
  uint32_t ack[kSrvSlots];  // device: number of the latest request served; written after rec / hit
  uint32_t hit[kSrvSlots];  // device: Traverse's bool
  uint32_t ticks[kSrvSlots]; // device: 10 ns ticks from the poll that found the request to its acknowledgement
  uint32_t stop;            // host: != 0 asks the running launch to leave (render entry points, scene destruction)
  uint32_t pad0_[15];
  uint32_t exited_epoch;    // device: epoch of the last launch all of whose waves have left
  uint32_t pad1_[15];
  alignas(16) double ray[kSrvSlots][6]; // host: org, dir (what Traverse reads of a Ray)
  MgpuIntersection rec[kSrvSlots];  // device
#ifdef MGPU_SRV_PROFILE
  uint32_t prof[kSrvSlots][4];      // device, diagnostic builds: ticks of the ray load, of the traversal; nodes, triangles
#endif
};
struct TraceServerCtl { // device memory, zeroed on the server's stream before every launch
  uint32_t quit, exited;
  unsigned long long last_work; // wall_clock64() of the last poll that found a request
};
hipError_t launch_trace_server(int cap, hipStream_t s, const DScene &sc, TraceMailbox *mb, TraceServerCtl *ctl, uint32_t epoch,
                               unsigned long long idle_ticks, unsigned long long life_ticks, unsigned long long max_polls,
                               uint32_t stage_nodes_bytes, uint32_t stage_tris_bytes); // != 0: the scene is copied into LDS
void launch_count_add(hipStream_t s, int32_t *count, size_t npix, int passes); // single pass: count[px] += 1 only
// pass accumulation from tile-major planes (k_render_sm with several passes): plane_stride = tiles * 192 floats, or tiles * 64 for
// `mono` planes (one float per pixel: scenes of grey materials, whose three channels are equal)
void launch_accumulate_tiled(hipStream_t s, const float *planes, size_t plane_stride, bool mono, int passes, size_t n_floats, int win_w,
                             float *image, int32_t *count, bool resume);
// tile_order[0..n) = tile indices by descending cost (256 log buckets); zeroes cost[]. One workgroup.
// z_classes > 0 (tiles_x * tiles_y == n_tiles): along the Z-order curve of the tile grid, stably split into that many cost classes
// (k_order_tiles_z: scenes whose BVH stays in HBM); 0: the 256-bucket counting sort
void launch_order_tiles(hipStream_t s, uint32_t *cost, uint32_t n_tiles, uint32_t *order, uint32_t tiles_x = 0, uint32_t tiles_y = 0, int z_classes = 0);
// slot-ordered DTri records + per-slot shading normals (9 doubles with face-varying normals, else the geometric normal)
void launch_scene_layout(hipStream_t s, const double *verts, const uint32_t *faces, const uint32_t *indices,
                         const uint32_t *matIDs, const double *fv_normals, size_t nf, DTri *tris, double *slot_normal);
// WNode records of the wide traversal (nn + 1 of them, the last is the super root)
void launch_wide_layout(hipStream_t s, const MgpuNode *nodes, size_t nn, WNode *out);
// Leaf hints (mgpu_device.hpp, leaf_hint_make): on / off, the smallest leaf that gets one, and what a split must save
#ifndef MGPU_LEAF_HINTS
#define MGPU_LEAF_HINTS 1 // LDS-resident scene: two sub-boxes per large leaf, tested once when the leaf's TRI work starts (kHintMinTris)
#endif
#ifndef MGPU_HINT_MIN
#define MGPU_HINT_MIN 4 // (8: 5.23-5.28, 6: 5.21-5.25, 4: 5.21 ms on C2; 12: 5.38; without hints 5.38-5.44)
#endif
#ifndef MGPU_HINT_WORTH
#define MGPU_HINT_WORTH 0.85 // (0.7: 5.32-5.35, 0.95: 5.21-5.24)
#endif
#ifndef MGPU_WIDE_STACK_LDS
#define MGPU_WIDE_STACK_LDS 8 // (6 / 5 measured in round 3 with the treelet grown into the freed LDS: within 2 % either way)
#endif
constexpr int kWideStackLds = MGPU_WIDE_STACK_LDS; // far-child stack entries per lane kept in LDS by the wide traversal (16 bytes each)
// ---- fast mode (mgpu_render_f32.hip): the scene in float --------------------------------------------------------------
struct alignas(16) FNode { // 32 bytes: box (rounded outward), a / b = children or leaf run (see k_layout_f32)
  float bmin[3], bmax[3];
  uint32_t a, b;
};
struct alignas(16) FTri { // 48 bytes: p0, e1, e2, material
  float v[9];
  uint32_t mat;
  uint32_t pad[2];
};
struct FScene {
  const FNode *nodes;
  const FTri *tris;
  const float *normals; // slot order: 9 floats (face-varying) or 3 (geometric)
  const float *diffuse; // 3 * nm
  uint32_t nm;
  int has_fv_normals;
  uint32_t *stack_overflow;
  uint32_t overflow_cap;
};
void launch_layout_f32(hipStream_t s, const MgpuNode *nodes, size_t nn, const DTri *tris, size_t nf, const double *slot_normal,
                       int has_fv, const double *mat_diffuse, uint32_t nm, FNode *fnodes, FTri *ftris, float *fnormals, float *fdiffuse);
hipError_t launch_render_f32(int cap, bool lds_scene, dim3 grid, hipStream_t s, size_t shmem, const FScene &sc, const RenderParams &p);
void launch_tonemap(hipStream_t s, const float *image, const int32_t *count, size_t npix, int mode, unsigned char *out);
constexpr size_t kLdsBudget = 160 * 1024 - 3072; // bytes of LDS per CU on gfx950, less the kernels' static part (launch parameters, cursor words, leaf tables, the sampler's azimuth table: up to 2.6 KB)

} // namespace mgpu
