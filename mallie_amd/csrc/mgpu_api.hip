// mgpu_api.hip -- host side of the C ABI declared in include/mgpu.h (and of the instrumentation in include/mgpu_internal.h).
//
// Scene upload re-lays the reference's Mesh + BVHAccel arrays out for the device (see mgpu_device.hpp: nodes verbatim,
// triangles pre-gathered per leaf slot with edges precomputed, shading normals per slot) and the entry points launch
// the gfx950 kernels of mgpu_kernels.hip.  There is no CPU execution path here: without a HIP device every call fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "mgpu_kernels.hpp"
#include "../../include/mgpu_internal.h"

using namespace mgpu;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                                             \
  do {                                                                                                            \
    hipError_t e_ = (expr);                                                                                       \
    if (e_ != hipSuccess) return fail(MGPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                                      __LINE__);                                                                  \
  } while (0)

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

constexpr int kCounterRing = 64;

} // namespace

// Scratch one render launch owns while it runs.  A slot is bound to the stream that used it last; a launch on another
// stream takes an unused slot or re-binds the least recently used one after waiting (on the device) for its last launch.
constexpr int kRenderSlots = 4;
constexpr size_t kTracePinnedRays = 4096; // mgpu_trace: batches up to this size use the pinned staging
// mgpu_trace, calls of up to kTraceCoalesceMax rays: concurrent callers are served together (see trace_coalesced)
constexpr size_t kTraceCoalesceMax = 64;
constexpr size_t kTraceZeroCopyRays = 1024; // rays per combined launch; they and their records live in host memory the GPU maps
struct TraceTicket {
  const MgpuRay *rays;
  size_t n;
  MgpuIntersection *out;
  uint8_t *hit;
  int rc = MGPU_OK;
  char err[256];
  std::atomic<int> done{0};
};
struct RenderSlot {
  hipStream_t stream = nullptr;
  bool used = false;
  unsigned long long last_use = 0;
  hipEvent_t done = nullptr;        // recorded after the slot's last launch
  float *p_planes = nullptr;        // per-pass radiance planes of k_render_sm (grow-only)
  size_t planes_floats = 0;
  uint32_t *p_tile_cost = nullptr;  // per 8x8 tile: cost of the last launch's pass 0 (k_render_sm), feeds k_order_tiles
  uint32_t *p_tile_order = nullptr; // hand-out order of the current launch
  size_t tile_cap = 0;
  long long tile_key[6] = {-1, -1, -1, -1, -1, -1}; // window/strip layout the costs belong to
  unsigned order_age = 0;           // launches of this layout so far (the hand-out order is renewed every few)
  void *p_overflow = nullptr;       // HBM stack overflow columns (deep trees only)
  size_t overflow_lanes = 0;
  void *p_woverflow = nullptr;      // the same for the wide traversal's far-child stack (16-byte entries)
  size_t woverflow_entries = 0;     // its capacity in entries (lanes x entries per lane of the launch that sized it)
  void *p_prim = nullptr;           // HBM-resident scene: staged primary rays, 40 bytes per lane of the launch (RenderParams::prim_stage)
  size_t prim_lanes = 0;
};

struct MgpuScene {
  int device = 0;
  size_t nv = 0, nf = 0, nn = 0, nm = 0;
  int tree_depth = 0;  // deepest node level (root = 0)
  uint32_t max_leaf_tris = 0; // largest leaf
  bool boxes_ordered = false; // bmin <= bmax in every reachable node (lets the kernels take the min/max slab test)
  int precision = MGPU_PRECISION_FP64; // mgpu_scene_set_precision: which render kernel family the render entry points use
  StreamScratch stream; // MGPU_RNG_STREAM: scratch of the chip-wide resolution, and the cached classification of a camera's pixels
  int stack_need = 1;  // entries a traversal can ever hold = tree_depth + 1
  int cap = 16;        // LDS stack entries per lane of the instantiated kernels
  double bmin[3], bmax[3];
  // leaf hints (mgpu_device.hpp, leaf_hint_make): centre and half diagonal of the box of the vertices the faces use -- measured from
  // the vertices, not taken from a caller's root node -- for scenes small enough to be rendered from LDS; hint_rho < 0: no hints
  double hint_c[3] = {0, 0, 0}, hint_rho = -1.0;
  DScene d{};
  DScene d_created{}; // `d` as mgpu_scene_create left it (overflow columns null): what never changes afterwards.  The trace server
                      // launches from a caller's thread without host_mutex and must not read fields ensure_overflow rewrites
  // owned device allocations
  void *p_nodes = nullptr, *p_tris = nullptr, *p_slotn = nullptr, *p_mat = nullptr, *p_verts = nullptr,
       *p_fnodes = nullptr, *p_ftris = nullptr, *p_fnormals = nullptr, *p_fdiffuse = nullptr, // fast mode (float copies)
       *p_faces = nullptr, *p_fvn = nullptr, *p_fvuv = nullptr, *p_overflow = nullptr, *p_wnodes = nullptr,
       *p_woverflow = nullptr, *p_treelet = nullptr, *p_treelet5 = nullptr;
  uint32_t treelet5_n = 0; // the smaller table k_render_w5 holds (two workgroups per CU share the LDS)
  size_t overflow_lanes = 0, woverflow_lanes = 0;
  uint32_t *p_counters = nullptr;         // kCounterRing work counters
  double stream_last_ms = 0.0;           // wall time of the last chip-wide resolution (all passes of the call)
  bool stream_last_fresh = false;        // ... and whether it had to classify the camera's pixels first
  unsigned long long stream_retries = 0; // MGPU_RNG_STREAM: attempts the chip-wide resolution had to repeat (a certain pixel that was not)
  unsigned long long *p_stats = nullptr;  // kStatWords
  unsigned launch_seq = 0;
  size_t device_bytes = 0;
  int num_cu = 0;
  int render_blocks_per_cu = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // per-launch kernel timing (HIP events on the launch stream), enabled by mgpu_timing_enable
  bool timing_on = false;
  std::vector<hipEvent_t> t_ev; // pairs: start, stop
  size_t t_used = 0;            // events used since the last mgpu_timing_read
  // Launch scratch of the render entry point, one set per stream in use so that frames on different streams overlap
  // (the end of one launch fills with the next frame's work; mallie_amd/frame.py keeps two frames in flight).
  RenderSlot slot[kRenderSlots];
  unsigned long long slot_clock = 0; // use counter for least-recently-used re-binding
  std::mutex host_mutex;            // serialises the host-buffer entry points (they share the staging below); the
                                    // reference calls Scene::Trace from all its OpenMP threads at once
  void *p_trace = nullptr;          // mgpu_trace: device staging of the host-buffer entry point (grow-only)
  size_t trace_cap = 0;             // rays it holds
  void *p_trace_pinned = nullptr;   // mgpu_trace, small batches: pinned host mirror of the staging (kTracePinnedRays rays)
  // submission queue of the single-ray callers (trace_coalesced)
  std::mutex q_mutex;
  std::condition_variable q_cv;
  std::vector<TraceTicket *> q;
  std::atomic<int> q_inside{0};  // host threads inside mgpu_trace's small-call path right now
  std::atomic<int> q_waiting{0}; // tickets queued and not yet taken by a leader
  bool q_leader = false;
  int q_expect = 1;              // callers the recent combined launches served: how many tickets a leader waits for (<= 20 us)
  void *p_trace_zc = nullptr;    // host memory mapped into the device: rays in, records + hit flags out (kTraceZeroCopyRays)
  unsigned long long q_batches = 0, q_tickets = 0; // combined launches and the calls they served (mgpu_trace_queue_stats)
  void *p_host_img = nullptr;       // mgpu_render: device landing buffer of the host-buffer entry point (grow-only)
  size_t host_img_bytes = 0;
  // mgpu_render with the render-ahead on (mgpu_scene_set_render_ahead): a progressive caller asks for the next passes of the same
  // camera call after call (render.cc's drivers do), and a call is a kernel FOLLOWED by 24.9 MB over PCIe (0.77 + 0.6 ms at 1080p,
  // one pass).  With the render-ahead, a call enqueues the frame the next call will most likely ask for -- same arguments, pass_base
  // moved on by `passes` -- on a stream of its own BEFORE it copies its own frame out, so that kernel runs under this copy; the
  // next call finds its frame (nearly) done.  A call that asks for anything else waits for the frame rendered ahead, drops it and
  // renders its own: same images either way (the frame depends on its arguments alone).
  struct AheadKey {
    double frame[12];
    float plane[4];
    int W, H, x0, y0, x1, y1, maxPathLength, passes, has_plane, precision;
    uint32_t pass_base;
    uint64_t seed;
  };
  bool ahead_on = false, ahead_valid = false, ahead_last_valid = false;
  AheadKey ahead_key;  // what the frame rendered ahead was rendered for
  AheadKey ahead_last; // the previous call's arguments: a frame is rendered ahead only for a caller seen to continue a sequence
  void *p_ahead[2] = {nullptr, nullptr}; // device frames: the one being copied out and the one rendered ahead
  size_t ahead_bytes = 0;
  int ahead_buf = 0;                     // which of the two holds the frame rendered ahead
  hipStream_t ahead_stream = nullptr;
  hipEvent_t ahead_done = nullptr;
  unsigned long long ahead_hits = 0, ahead_misses = 0;
  int last_slot = 0;                // slot of the last render launch (mgpu_debug_tile_order)
  unsigned long long *p_wave_log = nullptr; // 4 words x 16384 waves, diagnostic
  double *probe_buf = nullptr; // set only for the duration of mgpu_probe_path
  uint32_t probe_pixel = 0, probe_pass = 0;
  int pix_step = 1;            // set only for the duration of mgpu_render_step
  bool trace_queue_on = true;  // MGPU_TRACE_QUEUE=0: every small mgpu_trace call launches on its own (A/B measurements, tests)
  // resident trace server of the one-ray callers (trace_served; kernel: mgpu_trace_server.hip)
  bool srv_on = true;                 // MGPU_TRACE_SERVER=0: one-ray calls go through the submission queue instead
  bool srv_stage = true;              // MGPU_TRACE_SERVER_LDS=0: the server never copies the scene into LDS
  std::mutex srv_mutex;               // set-up, launches and retirements
  TraceMailbox *srv_mb = nullptr;     // host address of the mailbox (mapped, coherent)
  TraceMailbox *srv_mb_dev = nullptr; // the device's address of it
  TraceServerCtl *srv_ctl = nullptr;  // device
  void *srv_overflow = nullptr;       // the server's own stack overflow columns (deep trees only)
  hipStream_t srv_stream = nullptr;   // non-blocking: the default stream's work must not wait for a resident kernel
  std::atomic<uint32_t> srv_epoch{0}; // number of the latest launch (0: none yet); the mailbox says which one has ended
  std::atomic<bool> srv_ready{false};
  std::atomic<unsigned char> srv_busy[kSrvSlots]; // a caller owns a slot while its call lasts
  uint32_t srv_seq[kSrvSlots] = {};   // request numbers, written by the slot's owner
  std::atomic<unsigned long long> srv_launches{0}, srv_calls{0}, srv_ticks{0}; // ticks: device time of the served calls, 10 ns units
  unsigned long long srv_idle_us = 1000, srv_life_us = 100000; // MGPU_TRACE_SERVER_IDLE_US / _LIFE_US
  double srv_timeout_ms = 10000.0; // MGPU_TRACE_SERVER_TIMEOUT_MS: how long a caller waits for its record before the call fails
  bool tile_order_on = true;   // MGPU_TILE_ORDER (read once, when the scene is created)
  unsigned tile_order_every = 4; // MGPU_TILE_ORDER_EVERY
  int tile_order_z = INT_MIN;    // MGPU_TILE_ORDER_Z: forces the hand-out order of HBM-resident scenes (render_frames_impl); INT_MIN: by launch size
};

namespace {

int dev_alloc(MgpuScene *s, void **p, size_t bytes) {
  *p = nullptr;
  if (!bytes) return MGPU_OK;
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError(); // not sticky: the caller may retry with less
    return fail(MGPU_ERR_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  s->device_bytes += bytes;
  return MGPU_OK;
}

int upload(MgpuScene *s, void **p, const void *src, size_t bytes) {
  int rc = dev_alloc(s, p, bytes);
  if (rc) return rc;
  if (bytes) HIP_TRY(hipMemcpy(*p, src, bytes, hipMemcpyHostToDevice));
  return MGPU_OK;
}

// Validates the tree (child / leaf ranges in bounds, no cycles through an explicit visit budget) and returns its depth.
int tree_depth(const MgpuNode *nodes, size_t nn, size_t nf, int *depth_out, bool *boxes_ordered, uint32_t *max_leaf_out = nullptr) {
  struct Item { uint32_t node; int depth; };
  std::vector<Item> stack;
  stack.push_back({0u, 0});
  size_t visited = 0;
  int depth = 0;
  uint32_t max_leaf = 0;
  while (!stack.empty()) {
    Item it = stack.back();
    stack.pop_back();
    if (++visited > nn) return fail(MGPU_ERR_INVALID, "BVH is not a tree (more than %zu node visits)", nn);
    const MgpuNode &n = nodes[it.node];
    if (it.depth > depth) depth = it.depth;
    for (int k = 0; k < 3; k++) // false for NaNs too; an empty builder box (bmin = +max, bmax = -max) also lands here
      if (!(n.bmin[k] <= n.bmax[k])) *boxes_ordered = false;
    if (n.flag == 0) {
      if (n.axis < 0 || n.axis > 2) return fail(MGPU_ERR_INVALID, "node %u: bad axis %d", it.node, n.axis);
      for (int k = 0; k < 2; k++) {
        if (n.data[k] >= nn) return fail(MGPU_ERR_INVALID, "node %u: child %u out of range", it.node, n.data[k]);
        stack.push_back({n.data[k], it.depth + 1});
      }
    } else {
      if ((size_t)n.data[1] + n.data[0] > nf)
        return fail(MGPU_ERR_INVALID, "leaf %u: range [%u,+%u) exceeds %zu faces", it.node, n.data[1], n.data[0], nf);
      if (n.data[0] >= kWInterior) return fail(MGPU_ERR_INVALID, "leaf %u: %u triangles in one leaf", it.node, n.data[0]);
      if (n.data[0] > max_leaf) max_leaf = n.data[0];
    }
  }
  if (max_leaf_out) *max_leaf_out = max_leaf;
  *depth_out = depth;
  return MGPU_OK; // *boxes_ordered was initialised by the caller
}

// The treelet of the HBM-resident render kernel (mgpu_device.hpp, kWTreelet): wide records of the super root and of the
// interior nodes with the largest boxes -- a ray enters a node with a probability roughly proportional to its box's surface
// area -- taken parent before child, so the table is closed under "parent of".  The records are what k_wide_layout writes
// (boxes verbatim), except that references to children which are in the table too carry kWTreelet | their index there.
constexpr size_t kTreeletMaxRecords = (kLdsBudget - 16 * WStack<kWideStackLds>::kWaveBytes) / sizeof(WNode);
void build_treelet(const MgpuNode *nodes, size_t nn, size_t max_records, std::vector<WNode> &out) {
  out.clear();
  auto area = [&](uint32_t i) -> double {
    const MgpuNode &n = nodes[i];
    const double dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
    const double a = dx * dy + dy * dz + dz * dx;
    return a == a ? a : 0.0;
  };
  std::vector<uint32_t> picked; // node indices, in table order after the super root
  std::unordered_map<uint32_t, uint32_t> local; // node index -> table index
  std::priority_queue<std::pair<double, uint32_t>> heap;
  if (nodes[0].flag == 0) heap.push({area(0), 0u});
  while (!heap.empty() && picked.size() + 1 < max_records) {
    const uint32_t g = heap.top().second;
    heap.pop();
    local[g] = (uint32_t)picked.size() + 1u;
    picked.push_back(g);
    for (int k = 0; k < 2; ++k) {
      const uint32_t c = nodes[g].data[k];
      if (nodes[c].flag == 0) heap.push({area(c), c});
    }
  }
  auto child = [&](uint32_t c, double *box, uint32_t &ref, uint32_t &tag) {
    const MgpuNode &n = nodes[c];
    for (int k = 0; k < 3; ++k) {
      box[k] = n.bmin[k];
      box[3 + k] = n.bmax[k];
    }
    if (n.flag == 0) {
      const auto it = local.find(c);
      ref = it != local.end() ? (kWTreelet | it->second) : c;
      tag = kWInterior;
    } else {
      ref = n.data[1];
      tag = n.data[0];
    }
  };
  out.resize(picked.size() + 1);
  memset(out.data(), 0, out.size() * sizeof(WNode));
  child(0u, out[0].box0, out[0].ref0, out[0].tag0); // the super root, as k_wide_layout writes it
  for (int k = 0; k < 6; ++k) out[0].box1[k] = kDblMax;
  for (size_t i = 0; i < picked.size(); ++i) {
    WNode &w = out[i + 1];
    const MgpuNode &n = nodes[picked[i]];
    child(n.data[0], w.box0, w.ref0, w.tag0);
    child(n.data[1], w.box1, w.ref1, w.tag1);
    w.tag0 |= (uint32_t)n.axis << 30;
  }
}

int set_device(const MgpuScene *s) {
  HIP_TRY(hipSetDevice(s->device));
  return MGPU_OK;
}

// Makes sure the HBM overflow columns cover `lanes` hardware lanes.
int ensure_overflow(MgpuScene *s, size_t lanes) {
  const int extra = s->stack_need - s->cap;
  if (extra <= 0) {
    s->d.stack_overflow = nullptr;
    s->d.overflow_cap = 0;
    return MGPU_OK;
  }
  if (lanes > s->overflow_lanes) {
    if (s->p_overflow) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(s->p_overflow));
      s->p_overflow = nullptr;
    }
    int rc = dev_alloc(s, &s->p_overflow, lanes * (size_t)extra * sizeof(uint32_t));
    if (rc) return rc;
    s->overflow_lanes = lanes;
  }
  s->d.stack_overflow = (uint32_t *)s->p_overflow;
  s->d.overflow_cap = (uint32_t)extra;
  return MGPU_OK;
}

// The same for the wide traversal (k_trace_sm, HBM-resident k_render_env): far-child entries beyond the LDS part.
int ensure_woverflow(MgpuScene *s, size_t lanes) {
  const int extra = s->tree_depth - kWideStackLds; // at most one far child per interior level
  if (extra <= 0) {
    s->d.wstack_overflow = nullptr;
    s->d.woverflow_cap = 0;
    return MGPU_OK;
  }
  if (lanes > s->woverflow_lanes) {
    if (s->p_woverflow) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(s->p_woverflow));
      s->device_bytes -= s->woverflow_lanes * (size_t)extra * sizeof(uint4);
      s->p_woverflow = nullptr;
      s->woverflow_lanes = 0;
    }
    int rc = dev_alloc(s, &s->p_woverflow, lanes * (size_t)extra * sizeof(uint4));
    if (rc) return rc;
    s->woverflow_lanes = lanes;
  }
  s->d.wstack_overflow = (uint4 *)s->p_woverflow;
  s->d.woverflow_cap = (uint32_t)extra;
  return MGPU_OK;
}

// The slot a render launch on `st` uses (see RenderSlot).  Launches on one stream always find their own slot again, so
// the cost order of a repeated frame survives; the event wait makes re-binding safe without a host synchronisation.
int acquire_slot(MgpuScene *s, hipStream_t st, RenderSlot **out) {
  RenderSlot *pick = nullptr;
  for (RenderSlot &r : s->slot)
    if (r.used && r.stream == st) pick = &r;
  if (!pick)
    for (RenderSlot &r : s->slot)
      if (!r.used) { pick = &r; break; }
  if (!pick) {
    pick = &s->slot[0];
    for (RenderSlot &r : s->slot)
      if (r.last_use < pick->last_use) pick = &r;
    HIP_TRY(hipStreamWaitEvent(st, pick->done, 0));
  }
  if (!pick->done) HIP_TRY(hipEventCreateWithFlags(&pick->done, hipEventDisableTiming));
  pick->used = true;
  pick->stream = st;
  pick->last_use = ++s->slot_clock;
  s->last_slot = (int)(pick - s->slot);
  *out = pick;
  return MGPU_OK;
}

// Per-slot overflow columns for `lanes` hardware lanes; fills the launch's own copy of the scene descriptor.
int slot_overflow(MgpuScene *s, RenderSlot &r, size_t lanes, DScene &d) {
  const int extra = s->stack_need - s->cap;
  d.stack_overflow = nullptr;
  d.overflow_cap = 0;
  if (extra <= 0) return MGPU_OK;
  if (lanes > r.overflow_lanes) {
    if (r.p_overflow) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(r.p_overflow));
      s->device_bytes -= r.overflow_lanes * (size_t)extra * sizeof(uint32_t);
      r.p_overflow = nullptr;
      r.overflow_lanes = 0;
    }
    int rc = dev_alloc(s, &r.p_overflow, lanes * (size_t)extra * sizeof(uint32_t));
    if (rc) return rc;
    r.overflow_lanes = lanes;
  }
  d.stack_overflow = (uint32_t *)r.p_overflow;
  d.overflow_cap = (uint32_t)extra;
  return MGPU_OK;
}

// The same for the wide traversal of the HBM-resident render kernel.
int slot_woverflow(MgpuScene *s, RenderSlot &r, size_t lanes, DScene &d, int lds_entries = kWideStackLds) {
  const int extra = s->tree_depth - lds_entries;
  d.wstack_overflow = nullptr;
  d.woverflow_cap = 0;
  if (extra <= 0) return MGPU_OK;
  // `extra` differs between the kernels that share a slot (k_render_w5 keeps fewer entries in LDS than k_render_sm): the buffer is
  // sized, and its size remembered, in ENTRIES -- a column pitch of this launch times its lanes must fit, whatever launch allocated it
  const size_t need = lanes * (size_t)extra;
  if (need > r.woverflow_entries) {
    if (r.p_woverflow) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(r.p_woverflow));
      s->device_bytes -= r.woverflow_entries * sizeof(uint4);
      r.p_woverflow = nullptr;
      r.woverflow_entries = 0;
    }
    int rc = dev_alloc(s, &r.p_woverflow, need * sizeof(uint4));
    if (rc) return rc;
    r.woverflow_entries = need;
  }
  d.wstack_overflow = (uint4 *)r.p_woverflow;
  d.woverflow_cap = (uint32_t)extra;
  return MGPU_OK;
}

void read_stats(const unsigned long long *w, MgpuStats *st) {
  st->trace_calls = w[kStatTraceCalls];
  st->real_rays = w[kStatRays];
  st->nodes = w[kStatNodes];
  st->tris = w[kStatTris];
  st->paths = w[kStatPaths];
  st->stack_overflow = 0;
}

// ---- mgpu_trace for callers that bring ONE ray at a time from many threads ------------------------------------------------------
// The reference calls Scene::Trace per ray from every OpenMP thread (scene.cc:253-315, render.cc:403).  A device launch per
// call, serialised by a mutex, prices every ray at a launch + two copies + a synchronisation.  Instead the calls that are
// inside this function at the same time are combined: each caller queues a ticket; one of them becomes the leader, gives the
// others that are already on their way a moment (<= 20 us, and only when some are) to queue theirs, packs all queued rays
// into host memory the device maps, launches ONE traversal over them that writes the records straight back into that memory
// (no copy engine in the path: two DMA round trips cost more than the traversal of a handful of rays), waits, and hands every
// ticket its records.  A caller that is alone pays no waiting time.  Results are the records the per-call path returns: the
// same kernel traces the same rays, only their grouping into launches changes.
int trace_batch_zero_copy(MgpuScene *s, std::vector<TraceTicket *> &batch, size_t total) {
  std::lock_guard<std::mutex> host_lock(s->host_mutex); // the device staging and stream 0 are shared with the large-batch path
  int rc = set_device(s);
  if (rc) return rc;
  const size_t per_ray = sizeof(MgpuRay) + sizeof(MgpuIntersection) + 16;
  if (!s->p_trace_zc) {
    hipError_t e = hipHostMalloc(&s->p_trace_zc, kTraceZeroCopyRays * per_ray + 64, hipHostMallocMapped);
    if (e != hipSuccess) {
      s->p_trace_zc = nullptr;
      return fail(MGPU_ERR_OOM, "hipHostMalloc(trace queue): %s", hipGetErrorString(e));
    }
  }
  unsigned char *host = (unsigned char *)s->p_trace_zc;
  void *dev_base = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&dev_base, host, 0));
  unsigned char *dev = (unsigned char *)dev_base;
  const size_t out_bytes = sizeof(MgpuIntersection) * kTraceZeroCopyRays, hit_bytes = kTraceZeroCopyRays;
  MgpuRay *h_rays = (MgpuRay *)(host + out_bytes + hit_bytes);
  size_t off = 0;
  for (TraceTicket *t : batch) {
    memcpy(h_rays + off, t->rays, sizeof(MgpuRay) * t->n);
    off += t->n;
  }
  rc = mgpu_trace_device(s, (const MgpuRay *)(dev + out_bytes + hit_bytes), total, (MgpuIntersection *)dev, dev + out_bytes, nullptr, nullptr);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(nullptr));
  off = 0;
  for (TraceTicket *t : batch) {
    memcpy(t->out, host + sizeof(MgpuIntersection) * off, sizeof(MgpuIntersection) * t->n);
    memcpy(t->hit, host + out_bytes + off, t->n);
    off += t->n;
  }
  return MGPU_OK;
}

int trace_coalesced(MgpuScene *s, TraceTicket &t) {
  using clock = std::chrono::steady_clock;
  s->q_inside.fetch_add(1);
  std::unique_lock<std::mutex> lk(s->q_mutex);
  s->q.push_back(&t);
  s->q_waiting.fetch_add(1);
  while (!t.done.load(std::memory_order_acquire)) {
    if (s->q_leader) {
      // somebody else is serving a batch: spin briefly on our own flag (a batch takes tens of microseconds; a condition
      // variable's wake-up alone costs that much), then sleep
      lk.unlock();
      const auto t0 = clock::now();
      bool done = false;
      while (!(done = t.done.load(std::memory_order_acquire)) && clock::now() - t0 < std::chrono::microseconds(200)) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      lk.lock();
      if (done) break;
      if (s->q_leader && !t.done.load(std::memory_order_acquire)) s->q_cv.wait_for(lk, std::chrono::milliseconds(1));
      continue;
    }
    s->q_leader = true;
    // Callers that are inside this function and have not queued yet are about to; and callers that were served by the last
    // launches are, as a rule, on their way back with their next ray (a loop over Scene::Trace in every thread): wait until as
    // many tickets are queued as the recent launches served, 20 us at most.  A caller that has been alone waits for nobody.
    const int expect = s->q_expect;
    if (s->q_inside.load() > s->q_waiting.load() || s->q_waiting.load() < expect) {
      lk.unlock();
      const auto t0 = clock::now();
      while ((s->q_inside.load() > s->q_waiting.load() || s->q_waiting.load() < expect) &&
             clock::now() - t0 < std::chrono::microseconds(20)) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      lk.lock();
    }
    std::vector<TraceTicket *> batch;
    size_t total = 0, taken = 0;
    for (TraceTicket *q : s->q) {
      if (total + q->n > kTraceZeroCopyRays) break;
      batch.push_back(q);
      total += q->n;
      ++taken;
    }
    s->q.erase(s->q.begin(), s->q.begin() + (long)taken);
    s->q_waiting.fetch_sub((int)taken);
    s->q_batches += 1;
    s->q_tickets += taken;
    s->q_expect = (int)taken >= s->q_expect ? (int)taken : (s->q_expect + (int)taken) / 2; // follows the callers up at once, down by halves
    if (s->q_expect < 1) s->q_expect = 1;
    lk.unlock();
    const int rc = trace_batch_zero_copy(s, batch, total);
    for (TraceTicket *b : batch) {
      b->rc = rc;
      if (rc) snprintf(b->err, sizeof(b->err), "%s", g_err); // the failure text is thread-local to the leader
      b->done.store(1, std::memory_order_release); // `b` may be gone as soon as this is visible: nothing touches it afterwards
    }
    lk.lock();
    s->q_leader = false;
    s->q_cv.notify_all();
  }
  lk.unlock();
  s->q_inside.fetch_sub(1);
  if (t.rc) fail(t.rc, "%s", t.err);
  return t.rc;
}

// ---- mgpu_trace with ONE ray: the resident server (mgpu_trace_server.hip) -------------------------------------------------------
// The submission queue above shares a launch among the callers that happen to be inside mgpu_trace together; a caller that is
// alone still pays the launch + completion round trip (~22 us).  Here no call launches anything as long as a server launch is
// alive: the caller takes a mailbox slot, writes its ray and a request number into host memory the device maps, and spins on the
// acknowledgement the device writes next to the finished record.  A launch leaves by itself after srv_idle_us without requests
// (and at the latest after srv_life_us), so hipDeviceSynchronize / hipFree elsewhere in the process wait at most that long; the
// render entry points retire it before they launch, because their persistent kernels want every CU to themselves.
inline void cpu_relax() {
#if defined(__x86_64__)
  __builtin_ia32_pause();
#endif
}

std::mutex g_srv_mutex;
std::vector<MgpuScene *> g_srv_scenes; // scenes whose server has been set up (any device)
// Render entry points in progress per device.  A render call retires the live servers of its device (its persistent kernel wants
// every CU and most of the LDS) -- and a one-ray caller spinning in trace_served would start the next server at once, in front of
// the render kernel.  While a render call holds its device, server_launch starts nothing and new one-ray calls go
// through the submission queue.
constexpr int kMaxDevices = 64;
std::atomic<int> g_render_hold[kMaxDevices];
struct RenderHold {
  int dev;
  explicit RenderHold(int device) : dev(device >= 0 && device < kMaxDevices ? device : -1) {
    if (dev >= 0) g_render_hold[dev].fetch_add(1, std::memory_order_acq_rel);
  }
  ~RenderHold() {
    if (dev >= 0) g_render_hold[dev].fetch_sub(1, std::memory_order_acq_rel);
  }
  RenderHold(const RenderHold &) = delete;
  RenderHold &operator=(const RenderHold &) = delete;
};
bool render_held(int device) { return device >= 0 && device < kMaxDevices && g_render_hold[device].load(std::memory_order_acquire) > 0; }

int server_init_locked(MgpuScene *s) {
  if (s->srv_ready.load()) return MGPU_OK;
  int rc = set_device(s);
  if (rc) return rc;
  void *host = nullptr;
  hipError_t e = hipHostMalloc(&host, sizeof(TraceMailbox), hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) return fail(MGPU_ERR_OOM, "hipHostMalloc(trace mailbox): %s", hipGetErrorString(e));
  memset(host, 0, sizeof(TraceMailbox));
  void *dev = nullptr;
  e = hipHostGetDevicePointer(&dev, host, 0);
  if (e == hipSuccess) e = hipMalloc((void **)&s->srv_ctl, sizeof(TraceServerCtl));
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->srv_stream, hipStreamNonBlocking);
  const int extra = s->stack_need - s->cap;
  if (e == hipSuccess && extra > 0) e = hipMalloc(&s->srv_overflow, (size_t)kSrvWaves * 64 * (size_t)extra * sizeof(uint32_t));
  if (e != hipSuccess) {
    (void)hipHostFree(host);
    if (s->srv_ctl) (void)hipFree(s->srv_ctl);
    if (s->srv_stream) (void)hipStreamDestroy(s->srv_stream);
    s->srv_ctl = nullptr; s->srv_stream = nullptr;
    return fail(MGPU_ERR_HIP, "trace server set-up: %s", hipGetErrorString(e));
  }
  for (int i = 0; i < kSrvSlots; i++) s->srv_busy[i].store(0);
  s->srv_mb = (TraceMailbox *)host;
  s->srv_mb_dev = (TraceMailbox *)dev;
  return MGPU_OK;
}

int server_init(MgpuScene *s) {
  std::lock_guard<std::mutex> g(g_srv_mutex); // lock order everywhere: registry, then scene
  std::lock_guard<std::mutex> lk(s->srv_mutex);
  if (s->srv_ready.load()) return MGPU_OK;
  int rc = server_init_locked(s);
  if (rc) return rc;
  g_srv_scenes.push_back(s);
  s->srv_ready.store(true, std::memory_order_release);
  return MGPU_OK;
}

inline bool server_alive(const MgpuScene *s) {
  const uint32_t ep = s->srv_epoch.load(std::memory_order_acquire);
  return ep != 0 && __atomic_load_n(&s->srv_mb->exited_epoch, __ATOMIC_ACQUIRE) != ep;
}

// Launches server number seen_epoch + 1 unless somebody else already has.
int server_launch(MgpuScene *s, uint32_t seen_epoch) {
  std::lock_guard<std::mutex> lk(s->srv_mutex);
  if (s->srv_epoch.load() != seen_epoch) return MGPU_OK;
  if (render_held(s->device)) return MGPU_OK; // a render call is in progress on this device: the caller keeps waiting
  int rc = set_device(s);
  if (rc) return rc;
  DScene d = s->d_created; // the scene as it was created, with the server's own overflow columns
  const int extra = s->stack_need - s->cap;
  d.stack_overflow = extra > 0 ? (uint32_t *)s->srv_overflow : nullptr;
  d.overflow_cap = extra > 0 ? (uint32_t)extra : 0u;
  HIP_TRY(hipMemsetAsync(s->srv_ctl, 0, sizeof(TraceServerCtl), s->srv_stream));
  const unsigned long long ticks_per_us = 100; // wall_clock64(): the constant 100 MHz counter
  // a scene whose nodes and triangles fit beside the stacks is walked from LDS (MGPU_TRACE_SERVER_LDS=0: never)
  const size_t nodes_bytes = sizeof(MgpuNode) * s->nn, tris_bytes = sizeof(DTri) * s->nf;
  bool stage = s->srv_stage && nodes_bytes + tris_bytes + (size_t)s->cap * 256 + 1024 <= kLdsBudget;
  hipError_t le = launch_trace_server(s->cap, s->srv_stream, d, s->srv_mb_dev, s->srv_ctl, seen_epoch + 1, s->srv_idle_us * ticks_per_us,
                                      s->srv_life_us * ticks_per_us, s->srv_life_us * 20ull + 1000ull, stage ? (uint32_t)nodes_bytes : 0u,
                                      stage ? (uint32_t)tris_bytes : 0u);
  if (le != hipSuccess && stage) { // a device that grants less LDS than this build assumes: walk the scene from HBM instead
    (void)hipGetLastError();
    s->srv_stage = false;
    le = launch_trace_server(s->cap, s->srv_stream, d, s->srv_mb_dev, s->srv_ctl, seen_epoch + 1, s->srv_idle_us * ticks_per_us,
                             s->srv_life_us * ticks_per_us, s->srv_life_us * 20ull + 1000ull, 0u, 0u);
  }
  HIP_TRY(le);
  s->srv_launches.fetch_add(1);
  s->srv_epoch.store(seen_epoch + 1, std::memory_order_release);
  return MGPU_OK;
}

// Asks a live launch to leave and waits until it has (tens of microseconds).  Called before anything that wants the whole
// device (persistent render kernels) or frees what the server reads.
int server_retire(MgpuScene *s) {
  if (!s->srv_ready.load() || !server_alive(s)) return MGPU_OK;
  std::lock_guard<std::mutex> lk(s->srv_mutex);
  if (!server_alive(s)) return MGPU_OK;
  __atomic_store_n(&s->srv_mb->stop, 1u, __ATOMIC_RELEASE);
  const double t0 = now_ms();
  while (server_alive(s)) {
    cpu_relax();
    if (now_ms() - t0 > 5000.0) {
      __atomic_store_n(&s->srv_mb->stop, 0u, __ATOMIC_RELEASE);
      return fail(MGPU_ERR_HIP, "trace server did not leave within 5 s");
    }
  }
  __atomic_store_n(&s->srv_mb->stop, 0u, __ATOMIC_RELEASE);
  return MGPU_OK;
}

// Every live server on `device` leaves: the persistent render kernels size themselves for a whole device.
int servers_retire_device(int device) {
  std::lock_guard<std::mutex> g(g_srv_mutex);
  for (MgpuScene *o : g_srv_scenes)
    if (o->device == device) {
      int rc = server_retire(o);
      if (rc) return rc;
    }
  return MGPU_OK;
}

void server_destroy(MgpuScene *s) {
  if (!s->srv_ready.load()) return;
  {
    std::lock_guard<std::mutex> g(g_srv_mutex);
    for (size_t i = 0; i < g_srv_scenes.size(); i++)
      if (g_srv_scenes[i] == s) { g_srv_scenes.erase(g_srv_scenes.begin() + (long)i); break; }
  }
  (void)server_retire(s);
  (void)hipStreamSynchronize(s->srv_stream);
  (void)hipStreamDestroy(s->srv_stream);
  (void)hipFree(s->srv_ctl);
  if (s->srv_overflow) (void)hipFree(s->srv_overflow);
  (void)hipHostFree(s->srv_mb);
  s->srv_ready.store(false);
  s->srv_mb = nullptr;
}

int trace_coalesced(MgpuScene *s, TraceTicket &t);

int trace_served(MgpuScene *s, const MgpuRay *ray, MgpuIntersection *out, uint8_t *hit) {
  if (render_held(s->device)) { // a render call is in progress on this device: no server until it has enqueued its work
    TraceTicket t;
    t.rays = ray; t.n = 1; t.out = out; t.hit = hit;
    t.err[0] = 0;
    return trace_coalesced(s, t);
  }
  if (!s->srv_ready.load(std::memory_order_acquire)) {
    int rc = server_init(s);
    if (rc) return rc;
  }
  TraceMailbox *mb = s->srv_mb;
  // a slot of our own for the call: threads start from different waves' slots, so as many waves as callers work at once
  static std::atomic<unsigned> next_thread{0};
  thread_local unsigned my = next_thread.fetch_add(1);
  unsigned slot = 0;
  for (unsigned tries = 0;; ++tries) {
    const unsigned j = my + tries; // waves first, then lanes: all kSrvSlots slots in kSrvSlots tries
    slot = (j % kSrvWaves) * kSrvSlotsPerWave + (j / kSrvWaves) % kSrvSlotsPerWave;
    unsigned char expected = 0;
    if (s->srv_busy[slot].compare_exchange_strong(expected, 1, std::memory_order_acquire)) break;
    if (tries >= (unsigned)kSrvSlots) cpu_relax(); // more callers than slots: wait for one
  }
  const uint32_t seq = ++s->srv_seq[slot];
  memcpy(&mb->ray[slot][0], ray->org, sizeof(double) * 3);
  memcpy(&mb->ray[slot][3], ray->dir, sizeof(double) * 3);
  __atomic_store_n(&mb->req[slot], seq, __ATOMIC_RELEASE);
  int rc = MGPU_OK;
  const double t0 = now_ms();
  for (unsigned spins = 0;; ++spins) {
    if (__atomic_load_n(&mb->ack[slot], __ATOMIC_ACQUIRE) == seq) break;
    const uint32_t ep = s->srv_epoch.load(std::memory_order_acquire);
    if (ep == 0 || __atomic_load_n(&mb->exited_epoch, __ATOMIC_ACQUIRE) == ep) {
      // nobody is serving (no launch yet, or the last one left before it saw this request): one of the waiting callers launches
      if (__atomic_load_n(&mb->ack[slot], __ATOMIC_ACQUIRE) == seq) break; // served by the launch that has just left
      rc = server_launch(s, ep);
      if (rc) break;
    }
    cpu_relax();
    if ((spins & 0xFFFu) == 0xFFFu && now_ms() - t0 > s->srv_timeout_ms) {
      rc = fail(MGPU_ERR_HIP, "trace server did not answer within %.0f s", s->srv_timeout_ms * 1e-3);
      break;
    }
  }
  if (!rc) {
    memcpy(out, &mb->rec[slot], sizeof(MgpuIntersection));
    *hit = (uint8_t)mb->hit[slot];
    s->srv_calls.fetch_add(1, std::memory_order_relaxed);
    s->srv_ticks.fetch_add(mb->ticks[slot], std::memory_order_relaxed);
#ifdef MGPU_SRV_PROFILE
    {
      static std::atomic<unsigned long long> pr[4], pn;
      for (int k = 0; k < 4; k++) pr[k].fetch_add(mb->prof[slot][k]);
      const unsigned long long n = pn.fetch_add(1) + 1;
      if (n % 2000 == 0)
        fprintf(stderr, "srv profile after %llu calls: ray load %.2f us, traversal %.2f us (%.1f nodes + %.1f tris), whole call on the device %.2f us\n", n,
                0.01 * pr[0] / n, 0.01 * pr[1] / n, (double)pr[2] / n, (double)pr[3] / n, 0.01 * s->srv_ticks.load() / s->srv_calls.load());
    }
#endif
  }
  s->srv_busy[slot].store(0, std::memory_order_release);
  return rc;
}

} // namespace

extern "C" {

int mgpu_abi_version(void) { return MGPU_ABI_VERSION; }

int mgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *mgpu_last_error(void) { return g_err; }

const char *mgpu_status_string(int status) {
  switch (status) {
  case MGPU_OK: return "ok";
  case MGPU_ERR_INVALID: return "invalid argument";
  case MGPU_ERR_NO_DEVICE: return "no usable HIP device";
  case MGPU_ERR_OOM: return "out of memory";
  case MGPU_ERR_HIP: return "HIP runtime error";
  case MGPU_ERR_STACK: return "traversal stack deeper than the reference's 512 entries";
  case MGPU_ERR_UNSUPPORTED: return "unsupported mode";
  default: return "unknown status";
  }
}

void mgpu_hash_state(uint64_t seed, uint32_t pass, uint32_t pixel, uint32_t state[4]) {
  hash_state(seed, pass, pixel, state);
}

int mgpu_scene_create(const double *verts, size_t nv, const uint32_t *faces, size_t nf, const uint32_t *matIDs,
                      const double *fv_normals, const double *fv_uvs, const MgpuNode *nodes, size_t nn,
                      const uint32_t *indices, const double *mat_diffuse, size_t nm, int device, MgpuScene **out) {
  if (!out) return fail(MGPU_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!verts || !faces || !nodes || !indices || nv == 0 || nf == 0 || nn == 0)
    return fail(MGPU_ERR_INVALID, "verts/faces/nodes/indices must be non-empty");
  if (nm && !mat_diffuse) return fail(MGPU_ERR_INVALID, "mat_diffuse is NULL with nm = %zu", nm);
  if (nf > 0xFFFFFFF0ull || nn > 0xFFFFFFF0ull) return fail(MGPU_ERR_INVALID, "scene too large for 32-bit indices");
  const int ndev = mgpu_device_count();
  if (ndev <= 0) return fail(MGPU_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(MGPU_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, ndev - 1);
  for (size_t i = 0; i < 3 * nf; i++)
    if (faces[i] >= nv) return fail(MGPU_ERR_INVALID, "faces[%zu] = %u >= %zu vertices", i, faces[i], nv);
  for (size_t i = 0; i < nf; i++)
    if (indices[i] >= nf) return fail(MGPU_ERR_INVALID, "indices[%zu] = %u >= %zu faces", i, indices[i], nf);
  int depth = 0;
  bool boxes_ordered = true;
  uint32_t max_leaf = 0;
  int rc = tree_depth(nodes, nn, nf, &depth, &boxes_ordered, &max_leaf);
  if (rc) return rc;
  if (depth + 1 > 512) return fail(MGPU_ERR_STACK, "tree depth %d needs more than the reference's 512 stack entries", depth);

  MgpuScene *s = new (std::nothrow) MgpuScene();
  if (!s) return fail(MGPU_ERR_OOM, "host allocation failed");
  s->device = device;
  s->nv = nv; s->nf = nf; s->nn = nn; s->nm = nm;
  s->tree_depth = depth;
  s->max_leaf_tris = max_leaf;
  s->boxes_ordered = boxes_ordered;
  s->stack_need = depth + 1;
  s->cap = pick_stack_cap(s->stack_need);
  for (int k = 0; k < 3; k++) { s->bmin[k] = nodes[0].bmin[k]; s->bmax[k] = nodes[0].bmax[k]; }
  if (sizeof(MgpuNode) * nn + sizeof(DTri) * nf <= kLdsBudget) {
    double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    bool finite = true;
    for (size_t i = 0; i < 3 * nf; i++)
      for (int k = 0; k < 3; k++) {
        const double x = verts[3 * (size_t)faces[i] + k];
        finite = finite && std::isfinite(x);
        lo[k] = std::min(lo[k], x);
        hi[k] = std::max(hi[k], x);
      }
    // ... and only when no ray the render kernel makes can be longer than 1 + 2^-10: a bounce direction is cos / sin weighted sum of
    // an orthonormal pair and the shading normal (render.cc:271-339), the shading normal a convex combination of the face's three
    for (size_t i = 0; fv_normals && finite && i < 3 * nf; i++) {
      const double *n = fv_normals + 3 * i;
      finite = n[0] * n[0] + n[1] * n[1] + n[2] * n[2] <= 1.0 + 0x1p-10;
    }
    if (finite) {
      double r2 = 0.0;
      for (int k = 0; k < 3; k++) {
        s->hint_c[k] = 0.5 * lo[k] + 0.5 * hi[k];
        const double h = std::max(hi[k] - s->hint_c[k], s->hint_c[k] - lo[k]);
        r2 += h * h;
      }
      const double rho = std::sqrt(r2) * (1.0 + 0x1p-40);
      if (std::isfinite(rho)) s->hint_rho = rho;
    }
  }

#define TRY_OR_FREE(expr)        \
  do {                           \
    int rc_ = (expr);            \
    if (rc_) {                   \
      mgpu_scene_destroy(s);     \
      return rc_;                \
    }                            \
  } while (0)

  {
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) {
      delete s;
      return fail(MGPU_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    }
  }
  TRY_OR_FREE(upload(s, &s->p_nodes, nodes, sizeof(MgpuNode) * nn));
  TRY_OR_FREE(upload(s, &s->p_mat, mat_diffuse, sizeof(double) * 3 * nm));
  TRY_OR_FREE(upload(s, &s->p_verts, verts, sizeof(double) * 3 * nv));
  TRY_OR_FREE(upload(s, &s->p_faces, faces, sizeof(uint32_t) * 3 * nf));
  TRY_OR_FREE(upload(s, &s->p_fvn, fv_normals, fv_normals ? sizeof(double) * 9 * nf : 0));
  TRY_OR_FREE(upload(s, &s->p_fvuv, fv_uvs, fv_uvs ? sizeof(double) * 6 * nf : 0));
  // slot-ordered triangle records (p0, e1 = p1-p0, e2 = p2-p0 of bvh_accel.cc:606-607, face id, material id) and
  // per-slot shading normals are laid out on the device from the arrays just uploaded (k_scene_layout)
  {
    void *d_idx = nullptr, *d_mat = nullptr;
    int lrc = upload(s, &d_idx, indices, sizeof(uint32_t) * nf);
    if (!lrc && matIDs) lrc = upload(s, &d_mat, matIDs, sizeof(uint32_t) * nf);
    if (!lrc) lrc = dev_alloc(s, &s->p_tris, sizeof(DTri) * nf);
    if (!lrc) lrc = dev_alloc(s, &s->p_slotn, sizeof(double) * (fv_normals ? 9 : 3) * nf);
    if (!lrc) lrc = dev_alloc(s, &s->p_wnodes, sizeof(WNode) * (nn + 1));
    hipError_t e = hipSuccess;
    if (!lrc) {
      launch_wide_layout(0, (const MgpuNode *)s->p_nodes, nn, (WNode *)s->p_wnodes);
      launch_scene_layout(0, (const double *)s->p_verts, (const uint32_t *)s->p_faces, (const uint32_t *)d_idx,
                          (const uint32_t *)d_mat, (const double *)s->p_fvn, nf, (DTri *)s->p_tris, (double *)s->p_slotn);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    if (d_idx) { (void)hipFree(d_idx); s->device_bytes -= sizeof(uint32_t) * nf; }
    if (d_mat) { (void)hipFree(d_mat); s->device_bytes -= sizeof(uint32_t) * nf; }
    if (lrc) {
      mgpu_scene_destroy(s);
      return lrc;
    }
    if (e != hipSuccess) {
      mgpu_scene_destroy(s);
      return fail(MGPU_ERR_HIP, "scene layout kernel: %s", hipGetErrorString(e));
    }
  }
  TRY_OR_FREE(dev_alloc(s, (void **)&s->p_counters, sizeof(uint32_t) * kCounterRing * kShards));
  TRY_OR_FREE(dev_alloc(s, (void **)&s->p_stats, sizeof(unsigned long long) * kStatWords));
  {
    hipError_t e = hipMemset(s->p_stats, 0, sizeof(unsigned long long) * kStatWords);
    if (e == hipSuccess) e = hipEventCreate(&s->ev0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev1);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
      mgpu_scene_destroy(s);
      return fail(MGPU_ERR_HIP, "scene setup: %s", hipGetErrorString(e));
    }
    s->num_cu = prop.multiProcessorCount;
  }
  s->d.nodes = (const MgpuNode *)s->p_nodes;
  s->d.wnodes = (const WNode *)s->p_wnodes;
  s->d.wroot = (uint32_t)nn;
  s->d.treelet = nullptr;
  s->d.treelet_n = 0;
  if (nn < 0x80000000ull && nf < 0x80000000ull) { // the flag bit of a treelet reference must be free in every other reference
    size_t max_records = kTreeletMaxRecords;
    if (const char *e = getenv("MGPU_TREELET")) max_records = atoll(e) < 0 ? 0 : std::min<size_t>((size_t)atoll(e), kTreeletMaxRecords);
    if (max_records >= 1) {
      std::vector<WNode> tl;
      build_treelet(nodes, nn, max_records, tl);
      TRY_OR_FREE(upload(s, &s->p_treelet, tl.data(), tl.size() * sizeof(WNode)));
      s->d.treelet = (const WNode *)s->p_treelet;
      s->d.treelet_n = (uint32_t)tl.size();
      // k_render_w5 (mgpu_render_w5.hip): two 640-thread workgroups per CU, each with its own, smaller table
      const bool w5_small = getenv("MGPU_W5_BLOCK") && atoi(getenv("MGPU_W5_BLOCK")) == 320; // (experiment: four workgroups of five waves)
      const size_t per_wg = (size_t)160 * 1024 / (w5_small ? 4 : 2) - 4096, w5_waves = (w5_small ? 5 : 10) * render_w5_wave_bytes();
      const size_t w5_max = per_wg > w5_waves ? (per_wg - w5_waves) / sizeof(WNode) : 0;
      if (w5_max >= 1 && getenv("MGPU_W5") && atoi(getenv("MGPU_W5")) != 0) { // (opt-in while the kernel is an experiment)
        build_treelet(nodes, nn, std::min(max_records, w5_max), tl);
        TRY_OR_FREE(upload(s, &s->p_treelet5, tl.data(), tl.size() * sizeof(WNode)));
        s->treelet5_n = (uint32_t)tl.size();
      }
    }
  }
  s->d.wstack_overflow = nullptr;
  s->d.woverflow_cap = 0;
  s->d.tris = (const DTri *)s->p_tris;
  s->d.slot_normal = (const double *)s->p_slotn;
  s->d.mat_diffuse = (const double *)s->p_mat;
  s->d.nm = (uint32_t)nm;
  s->d.has_fv_normals = fv_normals ? 1 : 0;
  s->d.verts = (const double *)s->p_verts;
  s->d.faces = (const uint32_t *)s->p_faces;
  s->d.fv_normals = (const double *)s->p_fvn;
  s->d.fv_uvs = (const double *)s->p_fvuv;
  s->d.stack_overflow = nullptr;
  s->d.overflow_cap = 0;
  s->d.boxes_ordered = s->boxes_ordered ? 1 : 0;
  s->d.grey = 1;
  for (size_t i = 0; i < nm; i++)
    if (!(mat_diffuse[3 * i] == mat_diffuse[3 * i + 1] && mat_diffuse[3 * i + 1] == mat_diffuse[3 * i + 2])) s->d.grey = 0;
  if (const char *e = getenv("MGPU_GREY")) // 0: the general three-channel kernel even for grey scenes (A/B measurements, tests)
    if (atoi(e) == 0) s->d.grey = 0;
  if (const char *e = getenv("MGPU_PLAIN_SLABS")) // 0: literal slab test only (A/B measurements, tests)
    if (atoi(e) == 0) s->d.boxes_ordered = 0;
  s->d_created = s->d;
  if (const char *e = getenv("MGPU_TRACE_QUEUE")) s->trace_queue_on = atoi(e) != 0;
  if (const char *e = getenv("MGPU_TRACE_SERVER")) s->srv_on = atoi(e) != 0;
  if (const char *e = getenv("MGPU_TRACE_SERVER_LDS")) s->srv_stage = atoi(e) != 0;
  if (const char *e = getenv("MGPU_TRACE_SERVER_IDLE_US")) s->srv_idle_us = (unsigned long long)(atoll(e) < 1 ? 1 : atoll(e));
  if (const char *e = getenv("MGPU_TRACE_SERVER_TIMEOUT_MS")) s->srv_timeout_ms = atof(e) < 1.0 ? 1.0 : atof(e);
  if (const char *e = getenv("MGPU_TRACE_SERVER_LIFE_US")) s->srv_life_us = (unsigned long long)(atoll(e) < 100 ? 100 : atoll(e));
  if (const char *e = getenv("MGPU_TILE_ORDER")) s->tile_order_on = atoi(e) != 0;
  if (const char *e = getenv("MGPU_TILE_ORDER_Z")) s->tile_order_z = atoi(e);
  if (const char *e = getenv("MGPU_TILE_ORDER_EVERY")) s->tile_order_every = atoi(e) < 1 ? 1u : (unsigned)atoi(e);
  *out = s;
  return MGPU_OK;
#undef TRY_OR_FREE
}

int mgpu_scene_destroy(MgpuScene *s) {
  if (!s) return MGPU_OK;
  (void)hipSetDevice(s->device);
  server_destroy(s);
  (void)hipDeviceSynchronize();
  void *ptrs[] = {s->p_nodes, s->p_tris, s->p_slotn, s->p_mat, s->p_verts, s->p_faces, s->p_fvn, s->p_fvuv,
                  s->p_overflow, s->p_counters, s->p_stats, s->p_wave_log, s->p_host_img, s->p_trace, s->p_wnodes,
                  s->p_woverflow, s->p_fnodes, s->p_ftris, s->p_fnormals, s->p_fdiffuse, s->p_treelet, s->p_treelet5};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (s->p_trace_pinned) (void)hipHostFree(s->p_trace_pinned);
  if (s->p_trace_zc) (void)hipHostFree(s->p_trace_zc);
  for (void *p : s->p_ahead)
    if (p) (void)hipFree(p);
  if (s->ahead_done) (void)hipEventDestroy(s->ahead_done);
  if (s->ahead_stream) (void)hipStreamDestroy(s->ahead_stream);
  {
    StreamScratch &X = s->stream;
    void *sp[] = {X.cls, X.C, X.J, X.U, X.base, X.block_sum, X.F, X.uflag, X.Sarr, X.USx, X.totals, X.bad, X.table, X.state, X.jump};
    for (void *p : sp)
      if (p) (void)hipFree(p);
  }
  for (RenderSlot &r : s->slot) {
    void *rp[] = {r.p_planes, r.p_tile_cost, r.p_tile_order, r.p_overflow, r.p_woverflow, r.p_prim};
    for (void *p : rp)
      if (p) (void)hipFree(p);
    if (r.done) (void)hipEventDestroy(r.done);
  }
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  for (hipEvent_t e : s->t_ev) (void)hipEventDestroy(e);
  delete s;
  return MGPU_OK;
}

int mgpu_scene_set_precision(MgpuScene *s, int precision) {
  if (!s) return fail(MGPU_ERR_INVALID, "scene is NULL");
  if (precision != MGPU_PRECISION_FP64 && precision != MGPU_PRECISION_FP32) return fail(MGPU_ERR_INVALID, "bad precision %d", precision);
  if (precision == MGPU_PRECISION_FP32 && !s->p_fnodes) { // first use: the float copy of the scene
    int rc = set_device(s);
    if (rc) return rc;
    const int per = s->d.has_fv_normals ? 9 : 3;
    rc = dev_alloc(s, &s->p_fnodes, sizeof(FNode) * (s->nn ? s->nn : 1));
    if (!rc) rc = dev_alloc(s, &s->p_ftris, sizeof(FTri) * (s->nf ? s->nf : 1));
    if (!rc) rc = dev_alloc(s, &s->p_fnormals, sizeof(float) * per * (s->nf ? s->nf : 1));
    if (!rc) rc = dev_alloc(s, &s->p_fdiffuse, sizeof(float) * 3 * (s->d.nm ? s->d.nm : 1));
    if (rc) return rc;
    launch_layout_f32(nullptr, (const MgpuNode *)s->p_nodes, s->nn, (const DTri *)s->p_tris, s->nf, s->d.slot_normal, s->d.has_fv_normals,
                      s->d.mat_diffuse, s->d.nm, (FNode *)s->p_fnodes, (FTri *)s->p_ftris, (float *)s->p_fnormals, (float *)s->p_fdiffuse);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(nullptr));
  }
  s->precision = precision;
  return MGPU_OK;
}

int mgpu_scene_bbox(const MgpuScene *s, double bmin[3], double bmax[3]) {
  if (!s || !bmin || !bmax) return fail(MGPU_ERR_INVALID, "NULL argument");
  for (int k = 0; k < 3; k++) { bmin[k] = s->bmin[k]; bmax[k] = s->bmax[k]; }
  return MGPU_OK;
}

size_t mgpu_scene_device_bytes(const MgpuScene *s) { return s ? s->device_bytes : 0; }

int mgpu_scene_device(const MgpuScene *s) { return s ? s->device : -1; }

int mgpu_trace_device(MgpuScene *s, const MgpuRay *d_rays, size_t n, MgpuIntersection *d_out, uint8_t *d_hit, void *stream,
                      MgpuStats *stats) {
  if (!s) return fail(MGPU_ERR_INVALID, "scene is NULL");
  if (n && (!d_rays || !d_out || !d_hit)) return fail(MGPU_ERR_INVALID, "rays/out/hit must be non-NULL");
  if ((uintptr_t)d_out & 15u) return fail(MGPU_ERR_INVALID, "d_out must be 16-byte aligned");
  const double t0 = now_ms();
  int rc = set_device(s);
  if (rc) return rc;
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n == 0) return MGPU_OK;
  hipStream_t st = (hipStream_t)stream;
  // kernel choice: "sm" = persistent wave-scheduled traversal (k_trace_sm), "v1" = one ray per lane to completion
  // (k_trace; also used for batches too small to fill the persistent grid or too large for 32-bit ray indices)
  // From 262 144 rays up the choice is made on the device: k_trace_probe looks at the batch's coherence, both kernels are
  // enqueued and the one that was not chosen returns at its first instruction (no host round trip).
  bool use_sm = n >= 16384 && n < 0xF0000000ull;
  bool use_v1 = !use_sm;
  bool probe = n >= 262144 && n < 0xF0000000ull;
  if (const char *e = getenv("MGPU_TRACE_KERNEL")) {
    probe = false;
    if (!strcmp(e, "v1")) use_sm = false;
    else if (!strcmp(e, "sm")) use_sm = n < 0xF0000000ull;
    else if (!strcmp(e, "auto")) probe = n >= 64 * 128 && n < 0xF0000000ull;
    else return fail(MGPU_ERR_INVALID, "MGPU_TRACE_KERNEL=%s (expected v1|sm|auto)", e);
    use_v1 = !use_sm;
  }
  if (probe) use_sm = use_v1 = true;
  const size_t all_blocks = (n + kBlock - 1) / kBlock;
  size_t blocks_sm = all_blocks, blocks_v1 = all_blocks;
  size_t res_sm = (size_t)s->num_cu * 4, res_v1 = (size_t)s->num_cu * 8; // sm: 16 waves per CU; v1: grid-stride waves
  if (const char *e = getenv("MGPU_TRACE_BLOCKS_PER_CU")) res_sm = res_v1 = (size_t)s->num_cu * (size_t)(atoi(e) < 1 ? 1 : atoi(e));
  if (blocks_sm > res_sm) blocks_sm = res_sm;
  if (blocks_v1 > res_v1) blocks_v1 = res_v1;
  rc = ensure_overflow(s, (blocks_v1 > blocks_sm ? blocks_v1 : blocks_sm) * kBlock);
  if (rc) return rc;
  rc = ensure_woverflow(s, blocks_sm * kBlock);
  if (rc) return rc;
  uint32_t *counter = s->p_counters + (size_t)(s->launch_seq++ % kCounterRing) * kShards; // word 0: sm's cursor, 1: select
  uint32_t *select = probe ? counter + 1 : nullptr;
  if (use_sm) HIP_TRY(hipMemsetAsync(counter, 0, sizeof(uint32_t), st));
  if (stats) {
    HIP_TRY(hipMemsetAsync(s->p_stats, 0, sizeof(unsigned long long) * kStatWords, st));
    HIP_TRY(hipEventRecord(s->ev0, st));
  }
  if (probe) {
    launch_trace_probe(st, d_rays, n, select);
    HIP_TRY(hipGetLastError());
  }
  if (use_sm)
    HIP_TRY(launch_trace_sm(dim3((unsigned)blocks_sm), st, s->d, d_rays, (uint32_t)n, d_out, d_hit, counter, s->p_stats,
                            select));
  if (use_v1) {
    launch_trace(s->cap, dim3((unsigned)blocks_v1), st, s->d, d_rays, n, d_out, d_hit, s->p_stats, select);
    HIP_TRY(hipGetLastError());
  }
  if (stats) {
    HIP_TRY(hipEventRecord(s->ev1, st));
    unsigned long long w[kStatWords];
    HIP_TRY(hipMemcpyAsync(w, s->p_stats, sizeof(w), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    read_stats(w, stats);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    stats->kernel_ms = ms;
    stats->total_ms = now_ms() - t0;
  }
  return MGPU_OK;
}

int mgpu_trace(MgpuScene *s, const MgpuRay *rays, size_t n, MgpuIntersection *out, uint8_t *hit, MgpuStats *stats) {
  if (!s) return fail(MGPU_ERR_INVALID, "scene is NULL");
  if (n && (!rays || !out || !hit)) return fail(MGPU_ERR_INVALID, "rays/out/hit must be non-NULL");
  if (n == 1 && !stats && s->srv_on) return trace_served(s, rays, out, hit); // Scene::Trace / BVHAccel::Traverse: one ray per call
  if (n >= 1 && n <= kTraceCoalesceMax && !stats && s->trace_queue_on) { // a few rays per call, possibly from many threads
    TraceTicket t;
    t.rays = rays; t.n = n; t.out = out; t.hit = hit;
    t.err[0] = 0;
    return trace_coalesced(s, t);
  }
  std::lock_guard<std::mutex> host_lock(s->host_mutex);
  const double t0 = now_ms();
  int rc = set_device(s);
  if (rc) return rc;
  if (stats && s->ahead_valid) HIP_TRY(hipStreamSynchronize(s->ahead_stream)); // (shared counters, see render_frames_impl)
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n == 0) return MGPU_OK;
  // device staging of the host-buffer entry point, kept with the scene (grow-only): Scene::Trace / BVHAccel::Traverse
  // come through here one ray at a time, and three hipMalloc + hipFree pairs cost more than the trace itself
  const size_t per_ray = sizeof(MgpuRay) + sizeof(MgpuIntersection) + 16; // + hit byte, padded
  if (n > s->trace_cap) {
    if (s->p_trace) {
      (void)hipFree(s->p_trace);
      s->device_bytes -= s->trace_cap * per_ray;
      s->p_trace = nullptr;
      s->trace_cap = 0;
    }
    const size_t cap = n < 4096 ? 4096 : n;
    rc = dev_alloc(s, (void **)&s->p_trace, cap * per_ray);
    if (rc) return rc;
    s->trace_cap = cap;
  }
  // Small batches -- Scene::Trace / BVHAccel::Traverse come through here ONE ray at a time -- go through a pinned mirror of
  // the staging with two asynchronous copies and one synchronisation (records and hit flags leave the device as one block)
  // instead of three blocking copies from / to pageable memory.
  if (n <= kTracePinnedRays) {
    const size_t out_bytes = sizeof(MgpuIntersection) * n, hit_bytes = (n + 15) & ~(size_t)15, ray_bytes = sizeof(MgpuRay) * n;
    if (!s->p_trace_pinned) {
      hipError_t e = hipHostMalloc(&s->p_trace_pinned, kTracePinnedRays * per_ray + 64, hipHostMallocDefault);
      if (e != hipSuccess) {
        s->p_trace_pinned = nullptr;
        return fail(MGPU_ERR_OOM, "hipHostMalloc(trace staging): %s", hipGetErrorString(e));
      }
    }
    unsigned char *dev = (unsigned char *)s->p_trace, *pin = (unsigned char *)s->p_trace_pinned;
    MgpuIntersection *d_out_s = (MgpuIntersection *)dev;
    uint8_t *d_hit_s = dev + out_bytes;
    MgpuRay *d_rays_s = (MgpuRay *)(dev + out_bytes + hit_bytes); // 16-byte aligned: both sizes are multiples of 8 / 16
    memcpy(pin + out_bytes + hit_bytes, rays, ray_bytes);
    HIP_TRY(hipMemcpyAsync(d_rays_s, pin + out_bytes + hit_bytes, ray_bytes, hipMemcpyHostToDevice, nullptr));
    MgpuStats dev_stats_s;
    rc = mgpu_trace_device(s, d_rays_s, n, d_out_s, d_hit_s, nullptr, stats ? &dev_stats_s : nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(pin, dev, out_bytes + n, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    memcpy(out, pin, out_bytes);
    memcpy(hit, pin + out_bytes, n);
    if (stats) {
      *stats = dev_stats_s;
      stats->total_ms = now_ms() - t0;
    }
    return MGPU_OK;
  }
  MgpuIntersection *d_out = (MgpuIntersection *)s->p_trace; // 16-byte aligned (hipMalloc), records first
  MgpuRay *d_rays = (MgpuRay *)((unsigned char *)s->p_trace + s->trace_cap * sizeof(MgpuIntersection));
  uint8_t *d_hit = (uint8_t *)s->p_trace + s->trace_cap * (sizeof(MgpuIntersection) + sizeof(MgpuRay));
  auto cleanup = [&]() {};
#define TRY_T(expr)                                                                                   \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      cleanup();                                                                                      \
      return fail(MGPU_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                       \
    }                                                                                                 \
  } while (0)
  TRY_T(hipMemcpy(d_rays, rays, sizeof(MgpuRay) * n, hipMemcpyHostToDevice));
  MgpuStats dev_stats;
  // counters and kernel time only when asked for (they cost two events, a memset and a read-back per call); the
  // device-to-host copies below wait for the kernel either way
  rc = mgpu_trace_device(s, d_rays, n, d_out, d_hit, nullptr, stats ? &dev_stats : nullptr);
  if (rc) {
    cleanup();
    return rc;
  }
  TRY_T(hipMemcpy(out, d_out, sizeof(MgpuIntersection) * n, hipMemcpyDeviceToHost));
  TRY_T(hipMemcpy(hit, d_hit, n, hipMemcpyDeviceToHost));
  if (stats) {
    *stats = dev_stats;
    stats->total_ms = now_ms() - t0;
  }
  cleanup();
  return MGPU_OK;
#undef TRY_T
}

} // extern "C"

// mgpu_render_stream / mgpu_render_step / mgpu_probe_path compute in double whatever mgpu_scene_set_precision says: their
// results are defined against the reference's own numbers (its random stream, its block fill, its per-iteration record)
struct Fp64Only {
  MgpuScene *s;
  int saved;
  explicit Fp64Only(MgpuScene *scene) : s(scene), saved(scene ? scene->precision : 0) {
    if (s) s->precision = MGPU_PRECISION_FP64;
  }
  ~Fp64Only() {
    if (s) s->precision = saved;
  }
};

// n_frames consecutive frames of `passes` passes each (frame f: passes pass_base + f * passes ...) into d_images[f] /
// d_counts[f].  With several frames the passes of as many frames as the plane budget holds go into ONE persistent launch
// (the kernel sees n * passes passes; every frame's planes are then summed into its own image): the end of a launch, where
// waves run out of work one by one, is paid once per launch instead of once per frame -- what matters when a GPU renders
// an eighth of a frame.  The images are those of n single-frame calls, bit for bit.
static int render_frames_impl(MgpuScene *s, const double frame[12], int W, int H, int x0, int x1, int y_first, int strip_h,
                              int y_period, int n_rows, int maxPathLength, int passes, const float plane[4], int rng_mode,
                              const uint32_t *d_rng_states, uint64_t seed, uint32_t pass_base, int n_frames,
                              float *const *d_images, int32_t *const *d_counts, void *stream, MgpuStats *stats) {
  if (!s || !frame || !d_images || n_frames < 1) return fail(MGPU_ERR_INVALID, "scene/frame/images must be non-NULL, n_frames >= 1");
  for (int f = 0; f < n_frames; ++f)
    if (!d_images[f]) return fail(MGPU_ERR_INVALID, "image %d is NULL", f);
  const int pstep = s ? s->pix_step : 1; // > 1: the window is given in step x step blocks of the W x H frame
  if (W <= 0 || H <= 0 || x0 < 0 || (long long)x1 * pstep > W || x0 > x1) return fail(MGPU_ERR_INVALID, "bad window columns");
  if (strip_h <= 0 || y_period < strip_h || n_rows < 0 || y_first < 0) return fail(MGPU_ERR_INVALID, "bad strip layout");
  if (maxPathLength < 1 || passes < 1) return fail(MGPU_ERR_INVALID, "maxPathLength and passes must be >= 1");
  if ((uint64_t)W * (uint64_t)H > 0xFFFFFFFFull) return fail(MGPU_ERR_INVALID, "frame too large");
  if (rng_mode == MGPU_RNG_STREAM)
    return fail(MGPU_ERR_UNSUPPORTED,
                "the reference's serial RNG stream is a chain over the whole frame: use mgpu_render_stream (or a table of "
                "start states with MGPU_RNG_TABLE)");
  if (rng_mode != MGPU_RNG_TABLE && rng_mode != MGPU_RNG_HASH) return fail(MGPU_ERR_INVALID, "bad rng_mode %d", rng_mode);
  if (rng_mode == MGPU_RNG_TABLE && !d_rng_states) return fail(MGPU_ERR_INVALID, "MGPU_RNG_TABLE needs rng_states");
  if (n_rows > 0) {
    const int j = n_rows - 1;
    const long long ylast = (long long)y_first + (long long)(j / strip_h) * y_period + (j % strip_h);
    if (ylast * pstep >= H) return fail(MGPU_ERR_INVALID, "strip layout reaches row %lld of a %d-row frame", ylast * pstep, H);
  }
  const double t0 = now_ms();
  int rc = set_device(s);
  RenderHold hold(s->device); // no server is (re)launched on this device until this call has enqueued its work
  if (!rc) rc = servers_retire_device(s->device); // a resident trace server leaves first: this launch wants every CU
  if (rc) return rc;
  if (stats && s->ahead_valid && (hipStream_t)stream != s->ahead_stream) // a frame rendered ahead (mgpu_render) books into the same
    HIP_TRY(hipStreamSynchronize(s->ahead_stream));                       // counters: it leaves before a call that wants to read them
  if (stats) memset(stats, 0, sizeof(*stats));
  const int win_w = x1 - x0;
  if (win_w == 0 || n_rows == 0) return MGPU_OK;
  hipStream_t st = (hipStream_t)stream;

  // kernel choice: "sm_lds" (state machine, scene staged in LDS, one 1024-thread workgroup per CU) when the BVH fits
  // next to the stacks, else "sm" (state machine, scene through L1/L2); "v1" = the first, ray-synchronous kernel.
  const size_t scene_lds = sizeof(MgpuNode) * s->nn + sizeof(DTri) * s->nf;
  int kern = 2; // 0 = v1, 1 = sm, 2 = sm_lds
  if (const char *e = getenv("MGPU_RENDER_KERNEL")) {
    if (!strcmp(e, "v1")) kern = 0;
    else if (!strcmp(e, "sm")) kern = 1;
    else if (!strcmp(e, "sm_lds")) kern = 2;
    else return fail(MGPU_ERR_INVALID, "MGPU_RENDER_KERNEL=%s (expected v1|sm|sm_lds)", e);
  }
  const int se_bytes = lds_stack_entry_bytes(s->nn); // LDS-resident scene: node indices of 1 or 2 bytes, depth + 1 per lane
  if (kern == 2 && se_bytes > 2) kern = 1;
  int block = kern == 2 ? 1024 : kBlock;
  size_t shmem = (size_t)(block / 64) * s->cap * 64 * sizeof(uint32_t);
  bool prim = false; // LDS-resident scene: the items' primary rays staged in LDS when 40 KB more fit (k_render_sm, PRIM)
  size_t lds_hint_cap = 0;
  if (kern == 2) {
    shmem = lds_stack_bytes((size_t)(block / 64), (size_t)s->stack_need, (size_t)se_bytes);
    if (shmem + scene_lds > kLdsBudget) {
      kern = 1;
      block = kBlock;
      shmem = (size_t)(block / 64) * s->cap * 64 * sizeof(uint32_t);
    } else {
      shmem += scene_lds;
      prim = render_sm_prim_bytes() != 0 && shmem + render_sm_prim_bytes() <= kLdsBudget && !getenv("MGPU_NO_PRIM");
      if (prim) shmem += render_sm_prim_bytes();
      // ... and behind that the leaf hints (mgpu_render_sm.hip, kHintMinTris): one 96-byte record per leaf at most
      // (the record's number rides in the upper half of tri_end: slots and records of a scene that fits LDS stay far below 2^16)
      if (!getenv("MGPU_NO_HINTS") && s->nf < 0x10000u && s->hint_rho >= 0.0)
        lds_hint_cap = std::min<size_t>(std::min<size_t>((kLdsBudget - shmem) / (kHintFloats * 4), ((size_t)s->nn + 1) / 2), 0xFFFEu);
      shmem += lds_hint_cap * (kHintFloats * 4);
    }
  }
  // BVH in HBM: the wide traversal's far-child stacks, and -- one 1024-thread workgroup per CU instead of four of 256 -- the
  // treelet of the scene behind them (mgpu_device.hpp, kWTreelet)
  const bool treelet = kern == 1 && s->d.treelet != nullptr;
  if (treelet) block = 1024;
#ifdef MGPU_EXP_768
  if (treelet && getenv("MGPU_RENDER_BLOCK") && atoi(getenv("MGPU_RENDER_BLOCK")) == 768) block = 768;
#endif
  if (kern == 1) shmem = (size_t)(block / 64) * WStack<kWideStackLds>::kWaveBytes + (treelet ? (size_t)s->d.treelet_n * sizeof(WNode) : 0);
  int per_cu = kern == 2 || treelet ? 1 : (kern == 1 ? 4 : 2); // workgroups per CU: 16 waves per CU for the state-machine kernels
  // k_render_w5 (mgpu_render_w5.hip): the same walk with the state divided by hand for five waves per SIMD -- grey scenes whose
  // references fit the 12-byte stack entries (mgpu_device.hpp, WStackP), path lengths and windows that fit the packed cold words
  bool w5 = false;
  if (kern == 1 && treelet && s->p_treelet5 && s->d.grey && s->nf < (1u << 24) && s->nn < (1u << 30) && s->max_leaf_tris < 64u && maxPathLength <= 255 &&
      passes <= 0xFFFF && win_w <= 0xFFFF && n_rows <= 0xFFFF && s->precision != MGPU_PRECISION_FP32) {
    const char *e = getenv("MGPU_W5");
    w5 = e && atoi(e) != 0;
  }
  uint32_t w5_treelet_n = 0;
  if (w5) {
    const char *e = getenv("MGPU_W5_BLOCK");
    block = e && atoi(e) == 320 ? 320 : 640;
    per_cu = block == 640 ? 2 : 4;
    const size_t per_wg = (size_t)160 * 1024 / (size_t)per_cu - 4096, waves = (size_t)(block / 64) * render_w5_wave_bytes();
    w5_treelet_n = (uint32_t)std::min<size_t>(s->treelet5_n, per_wg > waves ? (per_wg - waves) / sizeof(WNode) : 0);
    if (w5_treelet_n < 1 || w5_treelet_n < s->treelet5_n) w5 = false; // (a prefix of the table is not a table: its references point past it)
    else shmem = waves + (size_t)w5_treelet_n * sizeof(WNode);
    if (!w5) { block = 1024; per_cu = 1; }
  }
  // fast mode (mgpu_scene_set_precision): k_render_f32 on the float copy of the scene -- 32-byte nodes and 48-byte
  // triangles, so scenes twice the size still fit in LDS beside the stacks
  bool f32_lds = false;
  if (s->precision == MGPU_PRECISION_FP32) {
    if (kern == 0) return fail(MGPU_ERR_UNSUPPORTED, "MGPU_RENDER_KERNEL=v1 has no fast mode");
    const size_t fscene_lds = sizeof(FNode) * s->nn + sizeof(FTri) * s->nf;
    const size_t stacks = (size_t)16 * s->cap * 64 * sizeof(uint32_t);
    f32_lds = s->cap <= 24 && s->stack_need <= s->cap && stacks + fscene_lds <= kLdsBudget && !getenv("MGPU_F32_HBM");
    kern = 3;
    block = f32_lds ? 1024 : kBlock;
    shmem = (size_t)(block / 64) * s->cap * 64 * sizeof(uint32_t) + (f32_lds ? fscene_lds : 0);
    per_cu = f32_lds ? 1 : (s->cap <= 24 ? 5 : 4); // HBM-resident: 20 waves per CU while their stacks fit (mgpu_render_f32.hip)
  }
  if (const char *e = getenv("MGPU_RENDER_BLOCKS_PER_CU")) per_cu = atoi(e) < 1 ? 1 : atoi(e);
  // persistent grid: as many workgroups as stay resident, capped by the work available
  const uint64_t tiles = (uint64_t)((win_w + 7) / 8) * (uint64_t)((n_rows + 7) / 8);
  uint64_t blocks = (uint64_t)s->num_cu * (uint64_t)per_cu;
  const uint64_t max_useful = (tiles * 64 + block - 1) / block;
  if (blocks > max_useful) blocks = max_useful;
  if (blocks < 1) blocks = 1;
  RenderSlot *slot = nullptr;
  rc = acquire_slot(s, st, &slot);
  if (rc) return rc;
  RenderSlot &R = *slot;
  DScene dsc = s->d; // this launch's scene descriptor: the overflow columns are the slot's
  rc = slot_overflow(s, R, blocks * block, dsc);
  if (rc) return rc;
  if (kern == 1) {
    rc = slot_woverflow(s, R, blocks * block, dsc, w5 ? render_w5_stack_entries() : kWideStackLds);
    if (rc) return rc;
  }
  // HBM-resident scene with the treelet kernel: the items' primary rays staged per wave in device memory (k_render_sm, PRIM; MGPU_NO_PRIM=1: off)
  if (kern == 1 && treelet && !w5 && !getenv("MGPU_NO_PRIM")) {
    const size_t lanes = blocks * (size_t)block;
    if (lanes > R.prim_lanes) {
      if (R.p_prim) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(R.p_prim));
        s->device_bytes -= R.prim_lanes * 40;
        R.p_prim = nullptr;
        R.prim_lanes = 0;
      }
      rc = dev_alloc(s, &R.p_prim, lanes * 40);
      if (rc) return rc;
      R.prim_lanes = lanes;
    }
    prim = true;
  }

  RenderParams P;
  memcpy(P.frame, frame, sizeof(P.frame));
  if (plane) memcpy(P.plane, plane, sizeof(P.plane));
  else memset(P.plane, 0, sizeof(P.plane));
  P.has_plane = plane ? 1 : 0;
  {
    // real3::normalize of the plane normal (common.h:48-56): sqrt and division are IEEE-exact on both sides
    double n[3] = {(double)P.plane[0], (double)P.plane[1], (double)P.plane[2]};
    const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (std::fabs(len) > 1.0e-6) {
      const double inv = 1.0 / len;
      n[0] *= inv; n[1] *= inv; n[2] *= inv;
    }
    memcpy(P.plane_n, n, sizeof(n));
  }
  P.W = W; P.H = H; P.x0 = x0; P.x1 = x1;
  P.y_first = y_first; P.strip_h = strip_h; P.y_period = y_period; P.n_rows = n_rows;
  P.maxPathLength = maxPathLength; P.passes = passes;
  P.pix_step = pstep;
  P.inv_len[0] = 0.0;
  for (int L = 1; L <= 16; ++L) P.inv_len[L] = 1.0 / (double)L;
  fill_tail_unit(P);
  if (pstep != 1 && kern == 0) return fail(MGPU_ERR_UNSUPPORTED, "MGPU_RENDER_KERNEL=v1 has no pixel step");
  P.rng_mode = rng_mode;
  P.rng_states = d_rng_states;
  P.seed = seed;
  P.pass_base = pass_base;
  P.image = d_images[0];
  P.count = d_counts ? d_counts[0] : nullptr;
  P.out = d_images[0];
  P.pass_stride = 0;
  const size_t n_floats = 3 * (size_t)n_rows * (size_t)win_w;
  // Passes are rendered `group` at a time so that the per-pass planes stay below a fixed budget (8 GiB of the 288 unless
  // MGPU_PLANES_MAX_MB says otherwise); k_accumulate_tiled carries the running float sum from one group to the next, so the
  // additions and their order are those of a single launch.
  if (tiles >= ((uint64_t)1 << 28)) return fail(MGPU_ERR_INVALID, "window too large: %llu tiles", (unsigned long long)tiles);
  int group = passes;
  int fpl = 1; // frames per launch
  // a scene of grey materials renders three equal channels (k_render_sm<GREY>): its planes hold ONE float per pixel and pass, and
  // k_accumulate_tiled writes the sum to the three channels -- the same float additions per channel, a third of the plane traffic
  const bool mono_planes = (kern == 1 || kern == 2) && dsc.grey != 0;
  size_t plane_floats = 0;
  if (kern != 0 && passes > 1) {
    // 8 GiB: the 64 passes of the 3840x2160 configuration (6.4 GB) fit one launch.  Every launch ends with a drain (its last
    // paths finishing on a mostly idle GPU), so fewer, longer launches are faster: at 1 GiB the C5 frame took seven launches and
    // 272.5 ms, now one and 258.6; C3's 64 passes two launches and 18.0 ms, now one and 17.4.
    size_t budget = (size_t)8 << 30;
    if (const char *e = getenv("MGPU_PLANES_MAX_MB")) budget = (size_t)(atoll(e) < 1 ? 1 : atoll(e)) << 20;
    plane_floats = (size_t)tiles * (mono_planes ? 64 : 192); // tile-major planes: 64 pixel slots per 8x8 tile (edge tiles padded)
    const size_t fit = budget / (plane_floats * sizeof(float));
    if ((size_t)group > fit) group = fit < 1 ? 1 : (int)fit;
    // the work cursor addresses (tile, pass) items with 28 bits per XCD part
    while (group > 1 && tiles * (uint64_t)group >= ((uint64_t)1 << 28)) group = (group + 1) / 2;
    if (n_frames > 1 && group == passes && fit >= 2 * (size_t)passes) { // whole frames share a launch
      fpl = (int)std::min<size_t>((size_t)n_frames, fit / (size_t)passes);
      if (const char *e = getenv("MGPU_FRAMES_PER_LAUNCH")) fpl = std::max(1, std::min(fpl, atoi(e)));
      while (fpl > 1 && tiles * (uint64_t)fpl * (uint64_t)passes >= ((uint64_t)1 << 28)) --fpl;
    }
    // The planes are grow-only scratch, one set per stream in use.  When the device cannot give what the budget allows (smaller
    // GPUs, several scenes or ranks on one device), fewer frames share a launch, then fewer passes a group, before the call
    // fails: same images, more launches.
    for (;;) {
      const size_t need = plane_floats * (size_t)group * (size_t)fpl;
      if (need <= R.planes_floats) break;
      if (R.p_planes) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(R.p_planes));
        s->device_bytes -= R.planes_floats * sizeof(float);
        R.p_planes = nullptr;
        R.planes_floats = 0;
      }
      size_t free_b = 0, total_b = 0;
      const bool known = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
      if (!known) (void)hipGetLastError();
      // leave a quarter of what is free to the caller's own buffers -- while there is something smaller to fall back to; the
      // smallest layout (one pass of one frame) is simply tried
      const bool can_shrink = fpl > 1 || group > 1;
      rc = (known && can_shrink && need * sizeof(float) > free_b - free_b / 4) ? MGPU_ERR_OOM : dev_alloc(s, (void **)&R.p_planes, need * sizeof(float));
      if (!rc) {
        R.planes_floats = need;
        break;
      }
      if (rc != MGPU_ERR_OOM || (fpl == 1 && group == 1)) return rc == MGPU_ERR_OOM ? fail(MGPU_ERR_OOM, "no device memory for one pass plane (%zu bytes)", plane_floats * sizeof(float)) : rc;
      if (fpl > 1) fpl = (fpl + 1) / 2;
      else group = (group + 1) / 2;
    }
    P.out = R.p_planes;
    P.pass_stride = plane_floats;
  }
  P.stats = s->p_stats;
  P.lds_nodes_bytes = (uint32_t)(sizeof(MgpuNode) * s->nn);
  P.lds_tris_bytes = (uint32_t)(sizeof(DTri) * s->nf);
  P.stack_cap = (uint32_t)s->stack_need;
  P.lds_hint_cap = kern == 2 ? (uint32_t)lds_hint_cap : 0u;
  P.hint_c[0] = P.hint_c[1] = P.hint_c[2] = P.hint_q2 = P.hint_q = 0.0;
  if (P.lds_hint_cap) {
    // Which rays consult the hints, and the |org - p0| |dir| the pads are sized for (mgpu_device.hpp, leaf_hint_make): origins no
    // farther from the centre of the scene's vertices than this camera or the scene's own half diagonal, whichever is larger.
    double q2 = 0.0;
    for (int k = 0; k < 3; k++) {
      P.hint_c[k] = s->hint_c[k];
      q2 += (frame[k] - s->hint_c[k]) * (frame[k] - s->hint_c[k]);
    }
    const double Q = std::max(std::sqrt(q2), s->hint_rho) * (1.0 + 0x1p-20); // the kernel's own |org - c|^2 rounds a few 2^-53 off
    P.hint_q2 = Q * Q;
    P.hint_q = Q * (1.0 + 0x1p-20);
    if (!std::isfinite(P.hint_q2) || !std::isfinite(s->hint_rho + P.hint_q)) P.lds_hint_cap = 0u;
    // the consultation runs in float (leaf_hint_apply): every origin that may consult lies within Q of c, so |origin| <= |c| + Q must
    // stay below 2^26 for its slab products to be finite (the boxes' side of that bound is leaf_hint_make's)
    const double cbig = std::max(std::fabs(P.hint_c[0]), std::max(std::fabs(P.hint_c[1]), std::fabs(P.hint_c[2])));
    if (!(cbig + P.hint_q < 0x1p26)) P.lds_hint_cap = 0u;
  }
  if (treelet) P.lds_nodes_bytes = (uint32_t)(sizeof(WNode) * s->d.treelet_n); // what the HBM-resident kernel stages into LDS
  if (w5) {
    P.lds_nodes_bytes = (uint32_t)(sizeof(WNode) * w5_treelet_n);
    dsc.treelet = (const WNode *)s->p_treelet5;
    dsc.treelet_n = w5_treelet_n;
  }
  FScene fsc{};
  if (kern == 3) {
    P.lds_nodes_bytes = (uint32_t)(sizeof(FNode) * s->nn);
    P.lds_tris_bytes = (uint32_t)(sizeof(FTri) * s->nf);
    fsc.nodes = (const FNode *)s->p_fnodes;
    fsc.tris = (const FTri *)s->p_ftris;
    fsc.normals = (const float *)s->p_fnormals;
    fsc.diffuse = (const float *)s->p_fdiffuse;
    fsc.nm = s->d.nm;
    fsc.has_fv_normals = s->d.has_fv_normals;
    fsc.stack_overflow = dsc.stack_overflow;
    fsc.overflow_cap = dsc.overflow_cap;
  }
  if (!s->p_wave_log && getenv("MGPU_WAVE_LOG")) {
    rc = dev_alloc(s, (void **)&s->p_wave_log, sizeof(unsigned long long) * 8 * 16384);
    if (rc) return rc;
  }
  P.wave_log = s->p_wave_log;
  P.prim_stage = (kern == 1 && prim) ? (unsigned char *)R.p_prim : nullptr;
  P.probe = s->probe_buf;
  P.probe_pixel = s->probe_pixel;

  // Cost-ordered hand-out: the tiles that were expensive in the previous launch of this layout go first, so the launch
  // drains on cheap paths (C2: 8.5 -> 8.4 ms on the whole frame, 2.07 -> 1.56 ms on an eighth of it; 1M-triangle grid
  // 7.8 -> 7.7 ms).  MGPU_TILE_ORDER=0 keeps the image order.
  P.tile_order = nullptr;
  P.tile_cost = nullptr;
  const bool use_order = kern != 0 && tiles >= 2 * blocks && s->tile_order_on;
  // BVH in HBM, a launch long enough that its end does not matter (>= 1024 work items per wave: the 3840 x 2160 x 64 spp frame of the
  // 10 M-triangle grid has 2 025, the 1080p frames 126 and 506): the order follows the Z-order curve of the tile grid
  // (k_order_tiles_z), so that what an XCD's waves work on together -- and its L2 holds -- is a compact image region: C5 254.3 ->
  // 239.9 ms; on the short launches the cost sort wins (C4 4.49 vs 4.98 ms along the curve, 4.54 with 8 cost classes).
  // MGPU_TILE_ORDER_Z overrides: 0 = the 256-bucket cost sort, 1 = the pure curve, n = n octave classes, -p = cheapest p % last.
  int z_classes = 0;
  if (kern == 1) z_classes = s->tile_order_z != INT_MIN ? s->tile_order_z : (tiles * (uint64_t)group * (uint64_t)fpl >= 1024ull * blocks * (uint64_t)(block / 64) ? 1 : 0);
  if (use_order) {
    if (tiles > R.tile_cap) {
      HIP_TRY(hipDeviceSynchronize());
      if (R.p_tile_cost) { HIP_TRY(hipFree(R.p_tile_cost)); R.p_tile_cost = nullptr; }
      if (R.p_tile_order) { HIP_TRY(hipFree(R.p_tile_order)); R.p_tile_order = nullptr; }
      s->device_bytes -= 2 * R.tile_cap * sizeof(uint32_t);
      R.tile_cap = 0;
      rc = dev_alloc(s, (void **)&R.p_tile_cost, tiles * sizeof(uint32_t));
      if (rc) return rc;
      rc = dev_alloc(s, (void **)&R.p_tile_order, tiles * sizeof(uint32_t));
      if (rc) return rc;
      R.tile_cap = tiles;
      R.tile_key[0] = -1;
    }
    const long long key[6] = {win_w, n_rows, x0, y_first, strip_h, y_period};
    if (memcmp(key, R.tile_key, sizeof(key)) != 0) {
      HIP_TRY(hipMemsetAsync(R.p_tile_cost, 0, tiles * sizeof(uint32_t), st)); // all-zero costs = image order
      memcpy(R.tile_key, key, sizeof(key));
      R.order_age = 0;
    }
    P.tile_order = R.p_tile_order;
  }
  // The order is renewed on the first two launches of a layout (image order, then the first measured costs) and every
  // fourth launch after that (MGPU_TILE_ORDER_EVERY): a frame's costs change slowly, and the sort is one workgroup's work
  // (35 us at 1080p) in front of the launch.  Costs are recorded only by the launch right before a renewal, so the
  // table always holds ONE launch's costs (no sums that could wrap) and the other launches skip the atomics.
  auto order_for_next_launch = [&]() -> int {
    if (!use_order) return MGPU_OK;
    const unsigned every = s->tile_order_every;
    auto renews = [&](unsigned age) { return age < 2 || age % every == 0; };
    if (renews(R.order_age)) {
      launch_order_tiles(st, R.p_tile_cost, (uint32_t)tiles, R.p_tile_order, (uint32_t)((win_w + 7) / 8), (uint32_t)((n_rows + 7) / 8), z_classes); // outside the kernel-time bracket
      HIP_TRY(hipGetLastError());
    }
    P.tile_cost = renews(R.order_age + 1) ? R.p_tile_cost : nullptr;
    ++R.order_age;
    return MGPU_OK;
  };
  if (n_frames == 1) { // (several frames: once per launch, below)
    rc = order_for_next_launch();
    if (rc) return rc;
  }
  if (stats) {
    HIP_TRY(hipMemsetAsync(s->p_stats, 0, sizeof(unsigned long long) * kStatWords, st));
    HIP_TRY(hipEventRecord(s->ev0, st));
  }
  hipEvent_t tev0 = nullptr, tev1 = nullptr;
  if (s->timing_on && !stats) {
    if (s->t_used + 2 > s->t_ev.size()) {
      if (s->t_ev.size() >= 2 * 4096) return fail(MGPU_ERR_INVALID, "timing ring full: call mgpu_timing_read");
      hipEvent_t a, b;
      HIP_TRY(hipEventCreate(&a));
      HIP_TRY(hipEventCreate(&b));
      s->t_ev.push_back(a);
      s->t_ev.push_back(b);
    }
    tev0 = s->t_ev[s->t_used];
    tev1 = s->t_ev[s->t_used + 1];
    s->t_used += 2;
    HIP_TRY(hipEventRecord(tev0, st));
  }
  auto launch_passes = [&](int n_passes, uint32_t first_pass) -> int { // one launch of the render kernel: passes first_pass ...
    P.passes = n_passes;
    P.pass_base = pass_base + first_pass;
    P.rng_states = d_rng_states ? d_rng_states + (size_t)first_pass * (size_t)W * (size_t)H * 4 : nullptr;
    P.probe_pass = s->probe_pass - first_pass; // wraps out of range for passes of other launches
    P.work_counter = s->p_counters + (size_t)(s->launch_seq++ % kCounterRing) * kShards;
    HIP_TRY(hipMemsetAsync(P.work_counter, 0, sizeof(uint32_t) * kShards, st));
    if (kern == 0) {
      launch_render(s->cap, dim3((unsigned)blocks), st, dsc, P);
      HIP_TRY(hipGetLastError());
    } else if (kern == 3) {
      HIP_TRY(launch_render_f32(s->cap, f32_lds, dim3((unsigned)blocks), st, shmem, fsc, P));
    } else if (w5) {
      HIP_TRY(launch_render_w5(block, dim3((unsigned)blocks), st, shmem, dsc, P));
    } else {
      HIP_TRY(launch_render_sm(se_bytes, kern == 2, prim, block, dim3((unsigned)blocks), st, shmem, dsc, P));
    }
    return MGPU_OK;
  };
  // what closes a frame rendered by launches of its own: the last group's planes into the image
  auto close_frame = [&](float *d_image, int32_t *d_count) -> int {
    if (kern == 0) return MGPU_OK;
    if (passes > 1) {
      const int last = passes - ((passes - 1) / group) * group;
      launch_accumulate_tiled(st, R.p_planes, plane_floats, mono_planes, last, n_floats, win_w, d_image, d_count, passes > group);
      HIP_TRY(hipGetLastError());
    } else if (d_count) {
      launch_count_add(st, d_count, n_floats / 3, 1); // single pass: the kernel wrote the image itself
      HIP_TRY(hipGetLastError());
    }
    return MGPU_OK;
  };
  for (int f0 = 0; f0 < n_frames; f0 += fpl) {
    const int nf = std::min(fpl, n_frames - f0);
    if (nf > 1) { // the passes of nf frames in one launch, then every frame's planes into its image
      rc = order_for_next_launch();
      if (rc) return rc;
      rc = launch_passes(nf * passes, (uint32_t)f0 * (uint32_t)passes);
      if (rc) return rc;
      for (int i = 0; i < nf; ++i) {
        launch_accumulate_tiled(st, R.p_planes + (size_t)i * (size_t)passes * plane_floats, plane_floats, mono_planes, passes, n_floats,
                                win_w, d_images[f0 + i], d_counts ? d_counts[f0 + i] : nullptr, false);
        HIP_TRY(hipGetLastError());
      }
      continue;
    }
    float *const d_image = d_images[f0];
    int32_t *const d_count = d_counts ? d_counts[f0] : nullptr;
    P.image = d_image;
    P.count = d_count;
    if (!P.pass_stride) P.out = d_image; // single pass: the kernel writes the image itself
    if (n_frames > 1) {
      rc = order_for_next_launch();
      if (rc) return rc;
    }
    for (int g0 = 0; g0 < passes; g0 += group) {
      if (use_order && g0 > 0 && P.tile_cost) { // only a launch that recorded costs has anything to sort by (a sort of the
        // zeroed table would put the later groups, and the following frames, back into image order)
        launch_order_tiles(st, R.p_tile_cost, (uint32_t)tiles, R.p_tile_order, (uint32_t)((win_w + 7) / 8), (uint32_t)((n_rows + 7) / 8), z_classes);
        HIP_TRY(hipGetLastError());
      }
      const int g = passes - g0 < group ? passes - g0 : group;
      rc = launch_passes(g, (uint32_t)f0 * (uint32_t)passes + (uint32_t)g0);
      if (rc) return rc;
      if (g0 + g < passes) { // not the last group: fold it into the image now, the planes are reused
        launch_accumulate_tiled(st, R.p_planes, plane_floats, mono_planes, g, n_floats, win_w, d_image, d_count, g0 > 0);
        HIP_TRY(hipGetLastError());
      }
    }
    if (n_frames > 1) {
      rc = close_frame(d_image, d_count);
      if (rc) return rc;
    }
  }
  // kernel_ms / the timing ring measure the dominant kernel alone (k_render or k_render_sm) when ONE frame is rendered and
  // its passes fit one group (the benchmarked case); with several groups the interleaved partial sums are inside the
  // bracket, with several frames their per-frame sums as well
  if (tev1) HIP_TRY(hipEventRecord(tev1, st));
  if (stats) HIP_TRY(hipEventRecord(s->ev1, st));
  if (n_frames == 1) {
    rc = close_frame(d_images[0], d_counts ? d_counts[0] : nullptr);
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(R.done, st));
  if (stats) {
    unsigned long long w[kStatWords];
    HIP_TRY(hipMemcpyAsync(w, s->p_stats, sizeof(w), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    read_stats(w, stats);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    stats->kernel_ms = ms;
    stats->total_ms = now_ms() - t0;
  }
  return MGPU_OK;
}

extern "C" {

int mgpu_render_strips_device(MgpuScene *s, const double frame[12], int W, int H, int x0, int x1, int y_first,
                              int strip_h, int y_period, int n_rows, int maxPathLength, int passes,
                              const float plane[4], int rng_mode, const uint32_t *d_rng_states, uint64_t seed,
                              uint32_t pass_base, float *d_image, int32_t *d_count, void *stream, MgpuStats *stats) {
  if (!d_image) return fail(MGPU_ERR_INVALID, "scene/frame/d_image must be non-NULL");
  float *const images[1] = {d_image};
  int32_t *const counts[1] = {d_count};
  return render_frames_impl(s, frame, W, H, x0, x1, y_first, strip_h, y_period, n_rows, maxPathLength, passes, plane, rng_mode,
                            d_rng_states, seed, pass_base, 1, images, d_count ? counts : nullptr, stream, stats);
}

int mgpu_render_frames_device(MgpuScene *s, const double frame[12], int W, int H, int x0, int x1, int y_first, int strip_h,
                              int y_period, int n_rows, int maxPathLength, int passes, const float plane[4], int rng_mode,
                              const uint32_t *d_rng_states, uint64_t seed, uint32_t pass_base, int n_frames,
                              float *const *d_images, int32_t *const *d_counts, void *stream, MgpuStats *stats) {
  return render_frames_impl(s, frame, W, H, x0, x1, y_first, strip_h, y_period, n_rows, maxPathLength, passes, plane, rng_mode,
                            d_rng_states, seed, pass_base, n_frames, d_images, d_counts, stream, stats);
}

int mgpu_render(MgpuScene *s, const double origin[3], const double corner[3], const double du[3], const double dv[3],
                int W, int H, int x0, int y0, int x1, int y1, int maxPathLength, int passes, const float plane[4],
                int rng_mode, const uint32_t *rng_states, uint64_t seed, uint32_t pass_base, float *image_out,
                int32_t *count_out, MgpuStats *stats) {
  if (!s || !origin || !corner || !du || !dv || !image_out) return fail(MGPU_ERR_INVALID, "NULL argument");
  if (W <= 0 || H <= 0 || x0 < 0 || y0 < 0 || x1 > W || y1 > H || x0 > x1 || y0 > y1)
    return fail(MGPU_ERR_INVALID, "bad window");
  if (passes < 1) return fail(MGPU_ERR_INVALID, "passes must be >= 1");
  std::lock_guard<std::mutex> host_lock(s->host_mutex);
  const double t0 = now_ms();
  int rc = set_device(s);
  if (rc) return rc;
  if (stats) memset(stats, 0, sizeof(*stats));
  const int ww = x1 - x0, wh = y1 - y0;
  if (ww == 0 || wh == 0) return MGPU_OK;
  double frame[12];
  memcpy(frame + 0, origin, 24);
  memcpy(frame + 3, corner, 24);
  memcpy(frame + 6, du, 24);
  memcpy(frame + 9, dv, 24);
  const size_t img_bytes = sizeof(float) * 3 * (size_t)ww * wh;
  if (s->ahead_on && !stats && rng_mode == MGPU_RNG_HASH && ww == W) { // (MgpuScene::AheadKey)
    MgpuScene::AheadKey key;
    memset(&key, 0, sizeof(key));
    memcpy(key.frame, frame, sizeof(frame));
    if (plane) memcpy(key.plane, plane, sizeof(key.plane));
    key.W = W; key.H = H; key.x0 = x0; key.y0 = y0; key.x1 = x1; key.y1 = y1;
    key.maxPathLength = maxPathLength; key.passes = passes; key.has_plane = plane ? 1 : 0; key.precision = s->precision;
    key.pass_base = pass_base; key.seed = seed;
    if (!s->ahead_stream) {
      HIP_TRY(hipStreamCreateWithFlags(&s->ahead_stream, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&s->ahead_done, hipEventDisableTiming));
    }
    if (img_bytes > s->ahead_bytes) {
      HIP_TRY(hipStreamSynchronize(s->ahead_stream));
      s->ahead_valid = false;
      for (void *&p : s->p_ahead) {
        if (p) {
          (void)hipFree(p);
          s->device_bytes -= s->ahead_bytes;
          p = nullptr;
        }
      }
      s->ahead_bytes = 0;
      for (void *&p : s->p_ahead) {
        rc = dev_alloc(s, &p, img_bytes);
        if (rc) return rc;
      }
      s->ahead_bytes = img_bytes;
    }
    auto enqueue = [&](uint32_t pb, void *dst) -> int {
      return mgpu_render_strips_device(s, frame, W, H, x0, x1, y0, wh, wh, wh, maxPathLength, passes, plane, rng_mode, nullptr, seed, pb,
                                       (float *)dst, nullptr, s->ahead_stream, nullptr);
    };
    int cur;
    if (s->ahead_valid && memcmp(&key, &s->ahead_key, sizeof(key)) == 0) { // the frame rendered ahead is the frame asked for
      cur = s->ahead_buf;
      s->ahead_hits++;
    } else { // nothing rendered ahead, or something else: it is dropped (its kernel has to leave the GPU first anyway)
      s->ahead_misses++;
      cur = s->ahead_valid ? 1 - s->ahead_buf : 0;
      rc = enqueue(pass_base, s->p_ahead[cur]);
      if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(s->ahead_done, s->ahead_stream));
    HIP_TRY(hipEventSynchronize(s->ahead_done)); // this call's frame is complete
    // The next call's frame, under this call's copy -- once the caller has been SEEN to come back for the next passes: this call
    // repeats the previous one's arguments with pass_base moved on by `passes`.  (A caller that renders one frame -- the console
    // driver, main_console.cc:70 -- or moves the camera every call never pays for a frame nobody asks for, nor waits for one
    // when the scene is destroyed.)  A speculative launch that fails (no memory for its planes, say) is dropped, not reported:
    // the frame this call was asked for is complete.
    s->ahead_valid = false;
    MgpuScene::AheadKey follows = s->ahead_last;
    follows.pass_base += (uint32_t)passes;
    const bool in_sequence = s->ahead_last_valid && memcmp(&key, &follows, sizeof(key)) == 0;
    s->ahead_last = key;
    s->ahead_last_valid = true;
    if (in_sequence && pass_base + (uint32_t)passes >= pass_base) {
      if (enqueue(pass_base + (uint32_t)passes, s->p_ahead[1 - cur]) == MGPU_OK) {
        s->ahead_key = key;
        s->ahead_key.pass_base = pass_base + (uint32_t)passes;
        s->ahead_buf = 1 - cur;
        s->ahead_valid = true;
      } else {
        (void)hipGetLastError();
      }
    }
    hipError_t ce = hipMemcpy(image_out + 3 * (size_t)y0 * W, s->p_ahead[cur], img_bytes, hipMemcpyDeviceToHost);
    if (ce != hipSuccess) return fail(MGPU_ERR_HIP, "hipMemcpy of the frame failed: %s", hipGetErrorString(ce));
    if (count_out)
      for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) count_out[(size_t)y * W + x] += passes;
    return MGPU_OK;
  }
  // the frame's landing buffer on the device is kept with the scene (grow-only): a progressive renderer calls this once
  // per pass with the same size, and hipMalloc + hipFree cost ~0.2 ms of a 7 ms frame
  if (img_bytes > s->host_img_bytes) {
    if (s->p_host_img) {
      (void)hipFree(s->p_host_img);
      s->device_bytes -= s->host_img_bytes;
      s->p_host_img = nullptr;
      s->host_img_bytes = 0;
    }
    rc = dev_alloc(s, (void **)&s->p_host_img, img_bytes);
    if (rc) return rc;
    s->host_img_bytes = img_bytes;
  }
  float *d_img = (float *)s->p_host_img;
  uint32_t *d_states = nullptr;
  auto cleanup = [&]() {
    if (d_states) (void)hipFree(d_states);
  };
#define TRY_R(expr)                                                                                   \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      cleanup();                                                                                      \
      return fail(MGPU_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                       \
    }                                                                                                 \
  } while (0)
  if (rng_mode == MGPU_RNG_TABLE) {
    if (!rng_states) {
      cleanup();
      return fail(MGPU_ERR_INVALID, "MGPU_RNG_TABLE needs rng_states");
    }
    const size_t bytes = (size_t)passes * W * H * 16;
    TRY_R(hipMalloc((void **)&d_states, bytes));
    TRY_R(hipMemcpy(d_states, rng_states, bytes, hipMemcpyHostToDevice));
  }
  MgpuStats local;
  rc = mgpu_render_strips_device(s, frame, W, H, x0, x1, y0, wh, wh, wh, maxPathLength, passes, plane, rng_mode,
                                 d_states, seed, pass_base, d_img, nullptr, nullptr, &local);
  if (rc) {
    cleanup();
    return rc;
  }
  if (ww == W) // whole rows: one contiguous copy (a strided one of the same bytes is ~0.1 ms slower at 1080p)
    TRY_R(hipMemcpy(image_out + 3 * (size_t)y0 * W, d_img, img_bytes, hipMemcpyDeviceToHost));
  else
    TRY_R(hipMemcpy2D(image_out + 3 * ((size_t)y0 * W + x0), sizeof(float) * 3 * (size_t)W, d_img,
                      sizeof(float) * 3 * (size_t)ww, sizeof(float) * 3 * (size_t)ww, (size_t)wh, hipMemcpyDeviceToHost));
  cleanup();
  if (count_out)
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) count_out[(size_t)y * W + x] += passes;
  if (stats) {
    *stats = local;
    stats->total_ms = now_ms() - t0;
  }
  return MGPU_OK;
#undef TRY_R
}

// Scratch of the chip-wide stream resolution for a W x H frame (grow-only), and whether the classification cached in it belongs
// to another camera (then the caller's first kernel classifies again).
static void stream_scratch_free(MgpuScene *s) {
  StreamScratch &X = s->stream;
  void *ptrs[] = {X.cls, X.C, X.J, X.U, X.base, X.block_sum, X.F, X.uflag, X.Sarr, X.USx, X.totals, X.bad};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  X = StreamScratch();
}
static int stream_scratch_for(MgpuScene *s, const StreamParams &P, bool *fresh) {
  StreamScratch &X = s->stream;
  const size_t npix = (size_t)P.W * (size_t)P.H;
  if (npix > X.npix_cap) {
    stream_scratch_free(s);
    const size_t sarr = stream_scratch_sarr_cap(npix);
#define SALLOC(field, bytes)                                                                     \
  do {                                                                                           \
    hipError_t e_ = hipMalloc((void **)&X.field, (bytes));                                       \
    if (e_ != hipSuccess) {                                                                      \
      (void)hipGetLastError();                                                                   \
      stream_scratch_free(s);                                                                    \
      return fail(MGPU_ERR_OOM, "hipMalloc(%zu) for the stream resolution failed", (size_t)(bytes)); \
    }                                                                                            \
  } while (0)
    SALLOC(cls, npix);
    SALLOC(C, 4 * npix);
    SALLOC(J, 4 * npix);
    SALLOC(U, 4 * npix);
    SALLOC(base, 16 * npix);
    SALLOC(block_sum, 8 * (npix / 1024 + 2));
    SALLOC(F, stream_scratch_f_bytes());
    SALLOC(uflag, npix);
    SALLOC(Sarr, 4 * sarr);
    SALLOC(USx, 4 * (npix + 1));
    SALLOC(totals, 8);
    SALLOC(bad, 8);
#undef SALLOC
    X.npix_cap = npix;
    X.sarr_cap = sarr;
  }
  const bool same = X.key_has_plane == P.has_plane && X.key_W == P.W && X.key_H == P.H && memcmp(X.key_frame, P.frame, sizeof(P.frame)) == 0 &&
                    memcmp(X.key_plane, P.plane, sizeof(P.plane)) == 0;
  *fresh = !same;
  if (!same) {
    memcpy(X.key_frame, P.frame, sizeof(P.frame));
    memcpy(X.key_plane, P.plane, sizeof(P.plane));
    X.key_has_plane = P.has_plane;
    X.key_W = P.W;
    X.key_H = P.H;
  }
  return MGPU_OK;
}

int mgpu_render_stream(MgpuScene *s, const double origin[3], const double corner[3], const double du[3], const double dv[3],
                       int W, int H, int maxPathLength, int passes, const float plane[4], uint32_t stream_state[4],
                       float *image_out, int32_t *count_out, uint32_t *states_out, MgpuStats *stats) {
  if (!s || !origin || !corner || !du || !dv || !image_out || !stream_state) return fail(MGPU_ERR_INVALID, "NULL argument");
  if (W <= 0 || H <= 0 || passes < 1 || maxPathLength < 1) return fail(MGPU_ERR_INVALID, "bad frame size / passes / maxPathLength");
  if (maxPathLength > kStreamMaxPathLength)
    return fail(MGPU_ERR_UNSUPPORTED, "MGPU_RNG_STREAM supports maxPathLength <= %d", kStreamMaxPathLength);
  if ((uint64_t)W * (uint64_t)H * (uint64_t)passes >= ((uint64_t)1 << 40)) return fail(MGPU_ERR_INVALID, "too many paths");
  const double t0 = now_ms();
  uint32_t *d_table = nullptr, *d_state = nullptr;
  uint4 *d_jump = nullptr;
  auto cleanup = [&]() {}; // (the three buffers live in the scene's stream scratch since round 6 and go with the scene)
#define TRY_S(expr)                                                             \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      cleanup();                                                                \
      return fail(MGPU_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    }                                                                           \
  } while (0)
  const size_t table_bytes = (size_t)passes * W * H * 16;
  // one lock over both phases (the state pass and the frame), and the caller's stream_state is advanced only when the frame
  // has been rendered: a failed call leaves the reference stream where it was
  std::lock_guard<std::mutex> host_lock(s->host_mutex);
  uint32_t next_state[4];
  int rc = set_device(s);
  RenderHold hold(s->device); // no server is (re)launched on this device until this call has enqueued its work
  if (!rc) rc = servers_retire_device(s->device); // a resident trace server leaves first: this launch wants every CU
  if (rc) return rc;
  {
    static std::vector<uint32_t> jump; // T^(2^j) over GF(2), computed once per process
    static std::mutex jump_mutex;
    {
      std::lock_guard<std::mutex> jl(jump_mutex);
      if (jump.empty()) {
        jump.resize((size_t)kStreamJumpBits * 128 * 4);
        stream_jump_matrices(jump.data());
      }
    }
    StreamScratch &X = s->stream;
    if (table_bytes > X.table_bytes) {
      if (X.table) {
        TRY_S(hipDeviceSynchronize());
        TRY_S(hipFree(X.table));
        X.table = nullptr;
        X.table_bytes = 0;
      }
      TRY_S(hipMalloc((void **)&X.table, table_bytes));
      X.table_bytes = table_bytes;
    }
    if (!X.state) TRY_S(hipMalloc((void **)&X.state, 16));
    if (!X.jump) {
      TRY_S(hipMalloc((void **)&X.jump, jump.size() * 4));
      hipError_t ej = hipMemcpy(X.jump, jump.data(), jump.size() * 4, hipMemcpyHostToDevice);
      if (ej != hipSuccess) {
        (void)hipFree(X.jump);
        X.jump = nullptr;
        return fail(MGPU_ERR_HIP, "upload of the jump matrices failed: %s", hipGetErrorString(ej));
      }
    }
    d_table = X.table;
    d_state = X.state;
    d_jump = X.jump;
    TRY_S(hipMemcpy(d_state, stream_state, 16, hipMemcpyHostToDevice));
    rc = ensure_overflow(s, 256);
    if (rc) {
      cleanup();
      return rc;
    }
    StreamParams P;
    memcpy(P.frame + 0, origin, 24);
    memcpy(P.frame + 3, corner, 24);
    memcpy(P.frame + 6, du, 24);
    memcpy(P.frame + 9, dv, 24);
    if (plane) memcpy(P.plane, plane, sizeof(P.plane));
    else memset(P.plane, 0, sizeof(P.plane));
    P.has_plane = plane ? 1 : 0;
    {
      double n[3] = {(double)P.plane[0], (double)P.plane[1], (double)P.plane[2]};
      const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      if (std::fabs(len) > 1.0e-6) {
        const double inv = 1.0 / len;
        n[0] *= inv; n[1] *= inv; n[2] *= inv;
      }
      memcpy(P.plane_n, n, sizeof(n));
    }
    P.W = W; P.H = H; P.maxPathLength = maxPathLength; P.passes = passes;
    P.jump = d_jump;
    P.state = d_state;
    P.table = d_table;
    const char *serial = getenv("MGPU_STREAM_SERIAL");
    if (serial && atoi(serial) != 0) { // round-3 kernel: one workgroup walks the chain (kept for A/B and as the tests' second opinion)
      TRY_S(launch_stream_states(s->cap, 0, s->d, P));
    } else {
      bool fresh = false;
      rc = stream_scratch_for(s, P, &fresh);
      if (!rc) rc = ensure_overflow(s, (size_t)s->num_cu * 4 * 256);
      if (rc) {
        cleanup();
        return rc;
      }
      uint32_t retries = 0;
      const double tr0 = now_ms();
      bool unsettled = false;
      hipError_t e = stream_states_resolve(s->cap, 0, s->d, P, s->stream, s->num_cu, fresh, &retries, &unsettled);
      s->stream_last_ms = now_ms() - tr0; // the resolution ends with a synchronisation (it reads the verification's verdict)
      s->stream_last_fresh = fresh;
      if (e == hipSuccess && unsettled) { // its verification kept finding new sub-pixel features (64 attempts): the one-workgroup walk needs no classes
        s->stream.key_has_plane = -1;
        TRY_S(hipMemcpy(d_state, stream_state, 16, hipMemcpyHostToDevice));
        TRY_S(launch_stream_states(s->cap, 0, s->d, P));
        retries = 64;
      }
      if (e != hipSuccess) {
        s->stream.key_has_plane = -1; // whatever is cached may be half-updated
        cleanup();
        return fail(MGPU_ERR_HIP, "stream_states_resolve failed: %s", hipGetErrorString(e));
      }
      s->stream_retries += retries;
    }
    TRY_S(hipMemcpy(next_state, d_state, 16, hipMemcpyDeviceToHost)); // waits for the kernels
  }
  // the frame itself: the ordinary renderer from that table, which stays on the device
  {
    const size_t img_bytes = sizeof(float) * 3 * (size_t)W * H;
    if (img_bytes > s->host_img_bytes) {
      if (s->p_host_img) {
        (void)hipFree(s->p_host_img);
        s->device_bytes -= s->host_img_bytes;
        s->p_host_img = nullptr;
        s->host_img_bytes = 0;
      }
      rc = dev_alloc(s, (void **)&s->p_host_img, img_bytes);
      if (rc) {
        cleanup();
        return rc;
      }
      s->host_img_bytes = img_bytes;
    }
    double frame[12];
    memcpy(frame + 0, origin, 24);
    memcpy(frame + 3, corner, 24);
    memcpy(frame + 6, du, 24);
    memcpy(frame + 9, dv, 24);
    MgpuStats local;
    Fp64Only fp64_guard(s);
    rc = mgpu_render_strips_device(s, frame, W, H, 0, W, 0, H, H, H, maxPathLength, passes, plane, MGPU_RNG_TABLE, d_table, 0, 0,
                                   (float *)s->p_host_img, nullptr, nullptr, &local);
    if (rc) {
      cleanup();
      return rc;
    }
    TRY_S(hipMemcpy(image_out, s->p_host_img, img_bytes, hipMemcpyDeviceToHost));
    if (states_out) TRY_S(hipMemcpy(states_out, d_table, table_bytes, hipMemcpyDeviceToHost));
    if (stats) *stats = local;
  }
  memcpy(stream_state, next_state, 16);
  cleanup();
  if (count_out)
    for (size_t i = 0; i < (size_t)W * H; i++) count_out[i] += passes;
  if (stats) stats->total_ms = now_ms() - t0;
  return MGPU_OK;
#undef TRY_S
}

int mgpu_scene_set_render_ahead(MgpuScene *s, int on) {
  if (!s) return fail(MGPU_ERR_INVALID, "scene is NULL");
  std::lock_guard<std::mutex> host_lock(s->host_mutex);
  if (!on && s->ahead_valid) { // whatever was rendered ahead leaves the GPU before the caller goes on
    int rc = set_device(s);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s->ahead_stream));
    s->ahead_valid = false;
  }
  s->ahead_on = on != 0;
  return MGPU_OK;
}

int mgpu_render_ahead_stats(MgpuScene *s, unsigned long long *hits, unsigned long long *misses) {
  if (!s) return fail(MGPU_ERR_INVALID, "scene is NULL");
  std::lock_guard<std::mutex> host_lock(s->host_mutex); // (a render call may be in the middle of updating them)
  if (hits) *hits = s->ahead_hits;
  if (misses) *misses = s->ahead_misses;
  return MGPU_OK;
}

int mgpu_debug_stream_classes(MgpuScene *s, unsigned char *out, size_t npix) { // diagnostic: the cached classification (0 / 1 / 2)
  if (!s || !out) return fail(MGPU_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> host_lock(s->host_mutex); // (mgpu_render_stream reallocates the scratch under this lock)
  if (!s->stream.cls || npix > s->stream.npix_cap) return fail(MGPU_ERR_INVALID, "no classification of that size is cached");
  int rc = set_device(s);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(out, s->stream.cls, npix, hipMemcpyDeviceToHost));
  return MGPU_OK;
}

int mgpu_stream_stats(MgpuScene *s, double *resolve_ms, int *classified, unsigned long long *retries, uint32_t *uncertain_pixels) {
  if (!s) return fail(MGPU_ERR_INVALID, "scene is NULL");
  std::lock_guard<std::mutex> host_lock(s->host_mutex); // (mgpu_render_stream reallocates the scratch under this lock)
  if (resolve_ms) *resolve_ms = s->stream_last_ms;
  if (classified) *classified = s->stream_last_fresh ? 1 : 0;
  if (retries) *retries = s->stream_retries;
  if (uncertain_pixels) {
    *uncertain_pixels = 0;
    if (s->stream.totals) {
      int rc = set_device(s);
      if (rc) return rc;
      HIP_TRY(hipMemcpy(uncertain_pixels, s->stream.totals, 4, hipMemcpyDeviceToHost));
    }
  }
  return MGPU_OK;
}

int mgpu_render_step(MgpuScene *s, const double origin[3], const double corner[3], const double du[3], const double dv[3],
                     int W, int H, int step, int maxPathLength, const float plane[4], int rng_mode,
                     const uint32_t *rng_states, uint64_t seed, uint32_t pass_base, float *image_out, int32_t *count_out,
                     MgpuStats *stats) {
  if (step == 1)
    return mgpu_render(s, origin, corner, du, dv, W, H, 0, 0, W, H, maxPathLength, 1, plane, rng_mode, rng_states, seed,
                       pass_base, image_out, count_out, stats);
  if (!s || !origin || !corner || !du || !dv || !image_out) return fail(MGPU_ERR_INVALID, "NULL argument");
  if (step < 1 || W <= 0 || H <= 0) return fail(MGPU_ERR_INVALID, "bad step / frame size");
  if (W % step || H % step)
    return fail(MGPU_ERR_UNSUPPORTED,
                "Render(step = %d) of a %dx%d frame: the reference's block fill (render.cc:684-696) writes outside the "
                "image unless both sizes are multiples of the step",
                step, W, H);
  std::lock_guard<std::mutex> host_lock(s->host_mutex);
  const double t0 = now_ms();
  int rc = set_device(s);
  if (rc) return rc;
  if (stats) memset(stats, 0, sizeof(*stats));
  const int cw = W / step, ch = H / step;
  double frame[12];
  memcpy(frame + 0, origin, 24);
  memcpy(frame + 3, corner, 24);
  memcpy(frame + 6, du, 24);
  memcpy(frame + 9, dv, 24);
  const size_t img_bytes = sizeof(float) * 3 * (size_t)cw * ch;
  if (img_bytes > s->host_img_bytes) {
    if (s->p_host_img) {
      (void)hipFree(s->p_host_img);
      s->device_bytes -= s->host_img_bytes;
      s->p_host_img = nullptr;
      s->host_img_bytes = 0;
    }
    rc = dev_alloc(s, (void **)&s->p_host_img, img_bytes);
    if (rc) return rc;
    s->host_img_bytes = img_bytes;
  }
  uint32_t *d_states = nullptr;
  if (rng_mode == MGPU_RNG_TABLE) {
    if (!rng_states) return fail(MGPU_ERR_INVALID, "MGPU_RNG_TABLE needs rng_states");
    const size_t bytes = (size_t)W * H * 16;
    HIP_TRY(hipMalloc((void **)&d_states, bytes));
    hipError_t e = hipMemcpy(d_states, rng_states, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(d_states);
      return fail(MGPU_ERR_HIP, "rng table upload: %s", hipGetErrorString(e));
    }
  }
  MgpuStats local;
  s->pix_step = step; // the window below counts step x step blocks; a block's path is its top-left pixel's
  Fp64Only fp64_guard(s);
  rc = mgpu_render_strips_device(s, frame, W, H, 0, cw, 0, ch, ch, ch, maxPathLength, 1, plane, rng_mode, d_states, seed,
                                 pass_base, (float *)s->p_host_img, nullptr, nullptr, &local);
  s->pix_step = 1;
  std::vector<float> coarse;
  if (!rc) {
    coarse.resize((size_t)3 * cw * ch);
    hipError_t e = hipMemcpy(coarse.data(), s->p_host_img, img_bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(MGPU_ERR_HIP, "read-back: %s", hipGetErrorString(e));
  }
  if (d_states) (void)hipFree(d_states);
  if (rc) return rc;
  // block fill, render.cc:684-696: every pixel of a block takes the block's radiance; count is incremented once per
  // colour channel there, i.e. by 3
  for (int by = 0; by < ch; by++)
    for (int bx = 0; bx < cw; bx++) {
      const float *src = &coarse[3 * ((size_t)by * cw + bx)];
      for (int v = 0; v < step; v++)
        for (int u = 0; u < step; u++) {
          const size_t px = (size_t)(by * step + v) * W + (size_t)(bx * step + u);
          image_out[3 * px + 0] = src[0];
          image_out[3 * px + 1] = src[1];
          image_out[3 * px + 2] = src[2];
          if (count_out) count_out[px] += 3;
        }
    }
  if (stats) {
    *stats = local;
    stats->total_ms = now_ms() - t0;
  }
  return MGPU_OK;
}

int mgpu_render_aov(MgpuScene *s, const double origin[3], const double corner[3], const double du[3], const double dv[3],
                    int W, int H, int kind, int rng_mode, const uint32_t *rng_states, uint64_t seed, uint32_t pass_base,
                    float *image_out, int32_t *count_out, MgpuStats *stats) {
  if (!s || !origin || !corner || !du || !dv || !image_out) return fail(MGPU_ERR_INVALID, "NULL argument");
  if (W <= 0 || H <= 0 || (uint64_t)W * (uint64_t)H > 0xFFFFFFFFull) return fail(MGPU_ERR_INVALID, "bad frame size");
  if (kind != MGPU_AOV_NORMAL && kind != MGPU_AOV_UV) return fail(MGPU_ERR_INVALID, "bad AOV kind %d", kind);
  if (rng_mode == MGPU_RNG_STREAM) return fail(MGPU_ERR_UNSUPPORTED, "MGPU_RNG_STREAM: use MGPU_RNG_TABLE with captured start states");
  if (rng_mode != MGPU_RNG_TABLE && rng_mode != MGPU_RNG_HASH) return fail(MGPU_ERR_INVALID, "bad rng_mode %d", rng_mode);
  if (rng_mode == MGPU_RNG_TABLE && !rng_states) return fail(MGPU_ERR_INVALID, "MGPU_RNG_TABLE needs rng_states");
  std::lock_guard<std::mutex> host_lock(s->host_mutex);
  const double t0 = now_ms();
  int rc = set_device(s);
  RenderHold hold(s->device); // no server is (re)launched on this device until this call has enqueued its work
  if (!rc) rc = servers_retire_device(s->device); // a resident trace server leaves first: this launch wants every CU
  if (rc) return rc;
  if (stats) memset(stats, 0, sizeof(*stats));
  const size_t npix = (size_t)W * H, img_bytes = sizeof(float) * 3 * npix;
  if (img_bytes > s->host_img_bytes) {
    if (s->p_host_img) {
      (void)hipFree(s->p_host_img);
      s->device_bytes -= s->host_img_bytes;
      s->p_host_img = nullptr;
      s->host_img_bytes = 0;
    }
    rc = dev_alloc(s, (void **)&s->p_host_img, img_bytes);
    if (rc) return rc;
    s->host_img_bytes = img_bytes;
  }
  uint32_t *d_states = nullptr;
  auto cleanup = [&]() {
    if (d_states) (void)hipFree(d_states);
  };
#define TRY_A(expr)                                                             \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      cleanup();                                                                \
      return fail(MGPU_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    }                                                                           \
  } while (0)
  if (rng_mode == MGPU_RNG_TABLE) {
    TRY_A(hipMalloc((void **)&d_states, npix * 16));
    TRY_A(hipMemcpy(d_states, rng_states, npix * 16, hipMemcpyHostToDevice));
  }
  size_t blocks = (npix + kBlock - 1) / kBlock;
  if (blocks > (size_t)s->num_cu * 8) blocks = (size_t)s->num_cu * 8;
  rc = ensure_overflow(s, blocks * kBlock);
  if (rc) {
    cleanup();
    return rc;
  }
  AovParams P;
  memcpy(P.frame + 0, origin, 24);
  memcpy(P.frame + 3, corner, 24);
  memcpy(P.frame + 6, du, 24);
  memcpy(P.frame + 9, dv, 24);
  P.W = W; P.H = H; P.mode = kind; P.rng_mode = rng_mode;
  P.rng_states = d_states;
  P.seed = seed;
  P.pass_base = pass_base;
  P.image = (float *)s->p_host_img;
  P.count = nullptr;
  P.stats = s->p_stats;
  TRY_A(hipMemsetAsync(s->p_stats, 0, sizeof(unsigned long long) * kStatWords, 0));
  TRY_A(hipEventRecord(s->ev0, 0));
  launch_render_aov(s->cap, dim3((unsigned)blocks), 0, s->d, P);
  TRY_A(hipGetLastError());
  TRY_A(hipEventRecord(s->ev1, 0));
  TRY_A(hipMemcpy(image_out, s->p_host_img, img_bytes, hipMemcpyDeviceToHost));
  unsigned long long w[kStatWords];
  TRY_A(hipMemcpy(w, s->p_stats, sizeof(w), hipMemcpyDeviceToHost));
  cleanup();
  if (count_out)
    for (size_t i = 0; i < npix; i++) count_out[i] += 1;
  if (stats) {
    read_stats(w, stats);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    stats->kernel_ms = ms;
    stats->total_ms = now_ms() - t0;
  }
  return MGPU_OK;
#undef TRY_A
}

int mgpu_render_panoramic_device(MgpuScene *s, const double origin[3], int W, int H, int x0, int y0, int x1, int y1,
                                 int maxPathLength, int samples, int stereo, int rng_mode, const uint32_t *d_rng_states,
                                 uint64_t seed, uint32_t pass_base, float *d_image, int32_t *d_count, void *stream,
                                 MgpuStats *stats) {
  if (!s || !origin || !d_image) return fail(MGPU_ERR_INVALID, "NULL argument");
  if (W <= 0 || H <= 0 || x0 < 0 || y0 < 0 || x1 > W || y1 > H || x0 > x1 || y0 > y1)
    return fail(MGPU_ERR_INVALID, "bad window");
  if (samples < 1 || maxPathLength < 1) return fail(MGPU_ERR_INVALID, "samples and maxPathLength must be >= 1");
  if (rng_mode == MGPU_RNG_STREAM)
    return fail(MGPU_ERR_UNSUPPORTED,
                "the reference's serial RNG stream cannot be reproduced in parallel; capture per-pixel start states "
                "and use MGPU_RNG_TABLE");
  if (rng_mode != MGPU_RNG_TABLE && rng_mode != MGPU_RNG_HASH) return fail(MGPU_ERR_INVALID, "bad rng_mode %d", rng_mode);
  if (rng_mode == MGPU_RNG_TABLE && !d_rng_states) return fail(MGPU_ERR_INVALID, "MGPU_RNG_TABLE needs rng_states");
  const double t0 = now_ms();
  int rc = set_device(s);
  RenderHold hold(s->device); // no server is (re)launched on this device until this call has enqueued its work
  if (!rc) rc = servers_retire_device(s->device); // a resident trace server leaves first: this launch wants every CU
  if (rc) return rc;
  if (stats) memset(stats, 0, sizeof(*stats));
  const int ww = x1 - x0, wh = y1 - y0;
  if (ww == 0 || wh == 0) return MGPU_OK;
  hipStream_t st = (hipStream_t)stream;
  const uint64_t tiles = (uint64_t)((ww + 7) / 8) * (uint64_t)((wh + 7) / 8);
  // scene placement as for mgpu_render: BVH in LDS (one 1024-thread workgroup per CU) when it fits beside the stacks
  const size_t scene_lds = sizeof(MgpuNode) * s->nn + sizeof(DTri) * s->nf;
  bool lds_scene = s->cap <= 24 && s->stack_need <= s->cap &&
                   (size_t)16 * s->cap * 64 * sizeof(uint32_t) + scene_lds <= kLdsBudget;
  if (const char *e = getenv("MGPU_RENDER_KERNEL")) {
    if (!strcmp(e, "sm")) lds_scene = false;
  }
  const int block = lds_scene ? 1024 : kBlock;
  uint64_t blocks = (uint64_t)s->num_cu * (lds_scene ? 1 : 4); // 16 waves per CU
  const uint64_t max_useful = (tiles * 64 + block - 1) / block;
  if (blocks > max_useful) blocks = max_useful;
  if (blocks < 1) blocks = 1;
  rc = ensure_overflow(s, blocks * block);
  if (rc) return rc;
  rc = ensure_woverflow(s, blocks * block);
  if (rc) return rc;
  EnvParams P;
  for (int L0 = 0; L0 <= 32; ++L0) { // EnvParams::tail_sum: the kernel's own loop, run here once per call
    volatile double r = 0.0;
    if (L0 >= 2 && L0 <= maxPathLength && maxPathLength <= 32)
      for (int L = L0; L <= maxPathLength; ++L) r = r + 0.5 / (double)(unsigned)L;
    P.tail_sum[L0] = r;
  }
  P.lds_nodes_bytes = (uint32_t)(sizeof(MgpuNode) * s->nn);
  P.lds_tris_bytes = (uint32_t)(sizeof(DTri) * s->nf);
  memcpy(P.origin, origin, sizeof(P.origin));
  {
    // psi = atan2(r, focal_length) with r = 0.5, focal_length = 4.0 (camera.cc:261-262,311): the host's libm, as the reference
    const double psi = std::atan2(0.5, 4.0);
    P.cos_psi = std::cos(psi);
    P.sin_psi = std::sin(psi);
  }
  P.W = W; P.H = H; P.x0 = x0; P.y0 = y0; P.win_w = ww; P.win_h = wh;
  P.maxPathLength = maxPathLength; P.samples = samples; P.stereo = stereo ? 1 : 0;
  P.rng_mode = rng_mode;
  P.rng_states = d_rng_states;
  P.seed = seed;
  P.pass_base = pass_base;
  P.image = d_image;
  P.count = d_count;
  P.work_counter = s->p_counters + (size_t)(s->launch_seq++ % kCounterRing) * kShards;
  P.stats = s->p_stats;
  HIP_TRY(hipMemsetAsync(P.work_counter, 0, sizeof(uint32_t), st));
  if (stats) {
    HIP_TRY(hipMemsetAsync(s->p_stats, 0, sizeof(unsigned long long) * kStatWords, st));
    HIP_TRY(hipEventRecord(s->ev0, st));
  }
  HIP_TRY(launch_render_env(s->cap, lds_scene, dim3((unsigned)blocks), st, s->d, P));
  if (stats) {
    HIP_TRY(hipEventRecord(s->ev1, st));
    unsigned long long w[kStatWords];
    HIP_TRY(hipMemcpyAsync(w, s->p_stats, sizeof(w), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    read_stats(w, stats);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    stats->kernel_ms = ms;
    stats->total_ms = now_ms() - t0;
  }
  return MGPU_OK;
}

int mgpu_render_panoramic(MgpuScene *s, const double origin[3], int W, int H, int x0, int y0, int x1, int y1,
                          int maxPathLength, int samples, int stereo, int rng_mode, const uint32_t *rng_states,
                          uint64_t seed, uint32_t pass_base, float *image_out, int32_t *count_out, MgpuStats *stats) {
  if (!s || !origin || !image_out) return fail(MGPU_ERR_INVALID, "NULL argument");
  if (W <= 0 || H <= 0 || x0 < 0 || y0 < 0 || x1 > W || y1 > H || x0 > x1 || y0 > y1)
    return fail(MGPU_ERR_INVALID, "bad window");
  if (samples < 1) return fail(MGPU_ERR_INVALID, "samples must be >= 1");
  std::lock_guard<std::mutex> host_lock(s->host_mutex);
  const double t0 = now_ms();
  int rc = set_device(s);
  if (rc) return rc;
  if (stats) memset(stats, 0, sizeof(*stats));
  const int ww = x1 - x0, wh = y1 - y0;
  if (ww == 0 || wh == 0) return MGPU_OK;
  float *d_img = nullptr;
  uint32_t *d_states = nullptr;
  auto cleanup = [&]() {
    if (d_img) (void)hipFree(d_img);
    if (d_states) (void)hipFree(d_states);
  };
#define TRY_R(expr)                                                                                   \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      cleanup();                                                                                      \
      return fail(MGPU_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                       \
    }                                                                                                 \
  } while (0)
  TRY_R(hipMalloc((void **)&d_img, sizeof(float) * 3 * (size_t)ww * wh));
  if (rng_mode == MGPU_RNG_TABLE) {
    if (!rng_states) {
      cleanup();
      return fail(MGPU_ERR_INVALID, "MGPU_RNG_TABLE needs rng_states");
    }
    const size_t bytes = (size_t)W * H * 16;
    TRY_R(hipMalloc((void **)&d_states, bytes));
    TRY_R(hipMemcpy(d_states, rng_states, bytes, hipMemcpyHostToDevice));
  }
  MgpuStats local;
  rc = mgpu_render_panoramic_device(s, origin, W, H, x0, y0, x1, y1, maxPathLength, samples, stereo, rng_mode, d_states,
                                    seed, pass_base, d_img, nullptr, nullptr, &local);
  if (rc) {
    cleanup();
    return rc;
  }
  TRY_R(hipMemcpy2D(image_out + 3 * ((size_t)y0 * W + x0), sizeof(float) * 3 * (size_t)W, d_img,
                    sizeof(float) * 3 * (size_t)ww, sizeof(float) * 3 * (size_t)ww, (size_t)wh, hipMemcpyDeviceToHost));
  cleanup();
  if (count_out)
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) count_out[(size_t)y * W + x] += samples;
  if (stats) {
    *stats = local;
    stats->total_ms = now_ms() - t0;
  }
  return MGPU_OK;
#undef TRY_R
}

int mgpu_trace_server_stats(MgpuScene *s, uint64_t *launches, uint64_t *calls, int *alive, double *device_us) {
  if (!s || !launches || !calls || !alive) return fail(MGPU_ERR_INVALID, "NULL argument");
  *launches = s->srv_launches.load();
  *calls = s->srv_calls.load();
  if (device_us) *device_us = *calls ? 0.01 * (double)s->srv_ticks.load() / (double)*calls : 0.0;
  *alive = s->srv_ready.load() && server_alive(s) ? 1 : 0;
  return MGPU_OK;
}

int mgpu_trace_server_retire(MgpuScene *s) {
  if (!s) return fail(MGPU_ERR_INVALID, "scene is NULL");
  return server_retire(s);
}

int mgpu_trace_calls_measure(MgpuScene *s, const MgpuRay *rays, size_t n, int threads, MgpuIntersection *out, uint8_t *hit,
                             double *calls_per_s) {
  if (!s || !rays || !out || !hit || !calls_per_s) return fail(MGPU_ERR_INVALID, "NULL argument");
  if (n == 0 || threads < 1 || threads > 1024) return fail(MGPU_ERR_INVALID, "n = %zu, threads = %d", n, threads);
  {
    uint8_t h0;
    MgpuIntersection i0;
    int rc = mgpu_trace(s, rays, 1, &i0, &h0, nullptr); // outside the clock: staging, the first server launch
    if (rc) return rc;
  }
  std::vector<int> rcs((size_t)threads, MGPU_OK);
  std::vector<std::string> errs((size_t)threads);
  std::vector<std::thread> pool;
  const auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < threads; t++)
    pool.emplace_back([&, t]() {
      for (size_t i = (size_t)t; i < n; i += (size_t)threads) {
        const int rc = mgpu_trace(s, rays + i, 1, out + i, hit + i, nullptr);
        if (rc) {
          rcs[(size_t)t] = rc;
          errs[(size_t)t] = g_err; // thread-local text of this worker
          return;
        }
      }
    });
  for (std::thread &t : pool) t.join();
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int t = 0; t < threads; t++)
    if (rcs[(size_t)t]) return fail(rcs[(size_t)t], "%s", errs[(size_t)t].c_str());
  *calls_per_s = (double)n / sec;
  return MGPU_OK;
}

int mgpu_trace_queue_stats(MgpuScene *s, uint64_t *launches, uint64_t *calls) {
  if (!s || !launches || !calls) return fail(MGPU_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> lk(s->q_mutex);
  *launches = s->q_batches;
  *calls = s->q_tickets;
  return MGPU_OK;
}

int mgpu_stats_read(MgpuScene *s, MgpuStats *out, int reset) {
  if (!s || !out) return fail(MGPU_ERR_INVALID, "NULL argument");
  int rc = set_device(s);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long w[kStatWords];
  HIP_TRY(hipMemcpy(w, s->p_stats, sizeof(w), hipMemcpyDeviceToHost));
  memset(out, 0, sizeof(*out));
  read_stats(w, out);
  if (reset) HIP_TRY(hipMemset(s->p_stats, 0, sizeof(w)));
  return MGPU_OK;
}

int mgpu_debug_words(MgpuScene *s, unsigned long long *out32) {
  if (!s || !out32) return fail(MGPU_ERR_INVALID, "NULL argument");
  int rc = set_device(s);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out32, s->p_stats, sizeof(unsigned long long) * 32, hipMemcpyDeviceToHost));
  return MGPU_OK;
}

int mgpu_occupancy_read(MgpuScene *s, MgpuOccupancy *out) {
  if (!s || !out) return fail(MGPU_ERR_INVALID, "NULL argument");
  int rc = set_device(s);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long w[kStatWords];
  HIP_TRY(hipMemcpy(w, s->p_stats, sizeof(w), hipMemcpyDeviceToHost));
  out->node_trips = w[kOccNodeTrips];
  out->node_lanes = w[kOccNodeLanes];
  out->tri_trips = w[kOccTriTrips];
  out->tri_lanes = w[kOccTriLanes];
  out->shade_steps = w[kOccShadeSteps];
  out->shade_lanes = w[kOccShadeLanes];
  out->node_steps = w[kOccNodeBooked];
  out->tri_steps = w[kOccTriBooked];
  out->sample_every = (uint32_t)kSampleEvery;
  out->pad_ = 0;
  return MGPU_OK;
}

int mgpu_debug_tile_order(MgpuScene *s, uint32_t *cost_out, uint32_t *order_out, size_t n_tiles) {
  if (!s) return fail(MGPU_ERR_INVALID, "NULL argument");
  const RenderSlot &R = s->slot[s->last_slot];
  if (!R.p_tile_cost || n_tiles > R.tile_cap) return fail(MGPU_ERR_INVALID, "no tile order for %zu tiles", n_tiles);
  int rc = set_device(s);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  if (cost_out) HIP_TRY(hipMemcpy(cost_out, R.p_tile_cost, sizeof(uint32_t) * n_tiles, hipMemcpyDeviceToHost));
  if (order_out) HIP_TRY(hipMemcpy(order_out, R.p_tile_order, sizeof(uint32_t) * n_tiles, hipMemcpyDeviceToHost));
  return MGPU_OK;
}

int mgpu_debug_wave_log(MgpuScene *s, unsigned long long *out, size_t n_waves) {
  if (!s || !out || !s->p_wave_log || n_waves > 16384) return fail(MGPU_ERR_INVALID, "no wave log");
  HIP_TRY(hipDeviceSynchronize());
  std::vector<unsigned long long> tmp(8 * 16384);
  HIP_TRY(hipMemcpy(tmp.data(), s->p_wave_log, sizeof(unsigned long long) * 8 * 16384, hipMemcpyDeviceToHost));
  for (size_t w = 0; w < n_waves; ++w) {
    memcpy(out + 8 * w, tmp.data() + 4 * w, 4 * sizeof(unsigned long long));
    memcpy(out + 8 * w + 4, tmp.data() + 4 * 16384 + 4 * w, 4 * sizeof(unsigned long long));
  }
  return MGPU_OK;
}

int mgpu_timing_enable(MgpuScene *s, int on) {
  if (!s) return fail(MGPU_ERR_INVALID, "scene is NULL");
  s->timing_on = on != 0;
  s->t_used = 0;
  return MGPU_OK;
}

int mgpu_timing_read(MgpuScene *s, double *total_ms, int *launches) {
  if (!s || !total_ms || !launches) return fail(MGPU_ERR_INVALID, "NULL argument");
  int rc = set_device(s);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  double sum = 0.0;
  for (size_t i = 0; i + 1 < s->t_used; i += 2) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->t_ev[i], s->t_ev[i + 1]));
    sum += ms;
  }
  *total_ms = sum;
  *launches = (int)(s->t_used / 2);
  s->t_used = 0;
  return MGPU_OK;
}

int mgpu_tonemap_device(int device, const float *d_image, const int32_t *d_count, size_t npix, int mode,
                        unsigned char *d_out, void *stream) {
  if (!d_image || !d_count || !d_out) return fail(MGPU_ERR_INVALID, "NULL argument");
  if (mode != MGPU_TONEMAP_LINEAR_RGB8 && mode != MGPU_TONEMAP_GAMMA22_BGRA8) return fail(MGPU_ERR_INVALID, "bad mode %d", mode);
  if (device < 0 || device >= mgpu_device_count()) return fail(MGPU_ERR_NO_DEVICE, "device %d not available", device);
  HIP_TRY(hipSetDevice(device));
  if (npix == 0) return MGPU_OK;
  launch_tonemap((hipStream_t)stream, d_image, d_count, npix, mode, d_out);
  HIP_TRY(hipGetLastError());
  return MGPU_OK;
}

int mgpu_probe_path(MgpuScene *s, const double frame[12], int W, int H, int px, int py, int maxPathLength,
                    const float plane[4], const uint32_t start_state[4], double *records, int *n_records) {
  if (!s || !frame || !start_state || !records || !n_records) return fail(MGPU_ERR_INVALID, "NULL argument");
  if (px < 0 || py < 0 || px >= W || py >= H || maxPathLength < 1) return fail(MGPU_ERR_INVALID, "bad pixel");
  int rc = set_device(s);
  if (rc) return rc;
  const size_t nrec = (size_t)maxPathLength * kProbeStride;
  double *d_probe = nullptr;
  float *d_img = nullptr;
  uint32_t *d_states = nullptr;
  auto cleanup = [&]() {
    s->probe_buf = nullptr;
    if (d_probe) (void)hipFree(d_probe);
    if (d_img) (void)hipFree(d_img);
    if (d_states) (void)hipFree(d_states);
  };
#define TRY_P(expr)                                                             \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      cleanup();                                                                \
      return fail(MGPU_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    }                                                                           \
  } while (0)
  TRY_P(hipMalloc((void **)&d_probe, sizeof(double) * nrec));
  TRY_P(hipMalloc((void **)&d_img, sizeof(float) * 3));
  // a one-pixel window reads its start state at table index (pass 0, pixel): place it there
  const size_t pix = (size_t)py * W + px;
  TRY_P(hipMalloc((void **)&d_states, 16 * (pix + 1)));
  TRY_P(hipMemcpy(d_states + 4 * pix, start_state, 16, hipMemcpyHostToDevice));
  std::vector<double> nanfill(nrec, std::nan(""));
  TRY_P(hipMemcpy(d_probe, nanfill.data(), sizeof(double) * nrec, hipMemcpyHostToDevice));
  s->probe_buf = d_probe;
  s->probe_pixel = (uint32_t)pix;
  s->probe_pass = 0;
  MgpuStats st;
  Fp64Only fp64_guard(s);
  rc = mgpu_render_strips_device(s, frame, W, H, px, px + 1, py, 1, 1, 1, maxPathLength, 1, plane, MGPU_RNG_TABLE,
                                 d_states, 0, 0, d_img, nullptr, nullptr, &st);
  if (rc) {
    cleanup();
    return rc;
  }
  std::vector<double> host(nrec);
  TRY_P(hipMemcpy(host.data(), d_probe, sizeof(double) * nrec, hipMemcpyDeviceToHost));
  int n = 0;
  while (n < maxPathLength && !std::isnan(host[(size_t)n * kProbeStride])) ++n;
  memcpy(records, host.data(), sizeof(double) * (size_t)n * kProbeStride);
  *n_records = n;
  cleanup();
  return MGPU_OK;
#undef TRY_P
}

} // extern "C"
