// mgpu_frame.hip -- multi-GPU frames behind the C ABI (include/mgpu.h, mgpu_frame_*): the image is cut into interleaved
// row strips, every GPU renders its strips with the scene replicated in its HBM, and ONE exchange step
// per frame brings the float RGB strips to rank 0 over RCCL / xGMI -- grouped ncclSend / ncclRecv of the strips straight
// into their final rows of rank 0's frame: no padding, no re-interleaving pass.
//
// Two ways to span the GPUs, same code:
//   * one process, n devices  (mgpu_frame_create: ncclCommInitAll; what mallie::Render uses with MALLIE_GPUS=n)
//   * one process per GPU     (mgpu_frame_create_rank: ncclCommInitRank with an id the caller distributes; bench.py under
//                              torch.distributed.run)
// RCCL is resolved at run time (dlopen "librccl.so.1": a process that already carries RCCL -- PyTorch's -- shares that
// copy), so single-GPU users never load it.  RCCL calls of a communicator are issued on ONE dedicated stream per device,
// in frame order; render streams hand over to it and take back with events, which is what lets several frames be in
// flight (the next frame's kernel runs under the previous frame's exchange and drain).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mgpu.h"
#include "mgpu_enqueue_pool.hpp"

// The handful of RCCL declarations this file needs, written out so that the library builds where the RCCL headers are not
// installed (single-GPU users never load RCCL at all): rccl.h, `ncclUniqueId` / `ncclResult_t` / `ncclDataType_t`.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;   // ncclSuccess = 0
typedef int ncclDataType_t; // ncclFloat32 = 7
}
constexpr ncclResult_t ncclSuccess = 0, ncclUnhandledCudaError = 1;
constexpr ncclDataType_t ncclFloat = 7;

namespace {

thread_local char g_ferr[512] = "";
int ffail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_ferr, sizeof(g_ferr), fmt, ap);
  va_end(ap);
  return code;
}

#define FHIP(expr)                                                                                                  \
  do {                                                                                                              \
    hipError_t e_ = (expr);                                                                                         \
    if (e_ != hipSuccess) return ffail(MGPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return MGPU_OK;
  void *h = nullptr;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return ffail(MGPU_ERR_UNSUPPORTED, "RCCL not found (librccl.so.1): %s", dlerror());
#define SYM(field, name)                                                                         \
  do {                                                                                           \
    *(void **)(&g_rccl.field) = dlsym(h, name);                                                  \
    if (!g_rccl.field) return ffail(MGPU_ERR_UNSUPPORTED, "RCCL symbol %s missing", name);       \
  } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommInitAll, "ncclCommInitAll");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(CommCount, "ncclCommCount");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.lib = h;
  return MGPU_OK;
}

#define FNCCL(expr)                                                                                               \
  do {                                                                                                            \
    ncclResult_t r_ = (expr);                                                                                     \
    if (r_ != ncclSuccess) return ffail(MGPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

constexpr int kMaxInFlight = 16;

// what one device keeps for one frame in flight
struct Slot {
  hipStream_t stream = nullptr; // render stream of this slot
  float *local = nullptr;       // this rank's strips, local rows contiguous (n_rows x W x 3)
  float *frame = nullptr;       // rank 0 only: the whole frame (H x W x 3)
  float *staging = nullptr;     // rank 0 only, block exchange: the other ranks' strip buffers as they arrive, rank after rank
  hipEvent_t rendered = nullptr, exchanged = nullptr;
  hipEvent_t x0 = nullptr, x1 = nullptr; // rank 0: the exchange step of this slot's frame on the communicator stream (timed)
  bool x_pending = false;                // x0 / x1 were recorded and not read yet
  // rank 0 with the read-back on (mgpu_frame_set_readback): the frame's landing buffer in pinned host memory and the event
  // behind its device-to-host copy on the member's read-back stream
  float *host = nullptr;
  hipEvent_t copied = nullptr;
  bool copy_wanted = false;  // a frame was enqueued with the read-back on and its copy has not been issued yet
  bool copy_pending = false; // the copy was issued: `copied` is behind it
};

struct Member { // one GPU of this process
  int rank = 0, device = 0;
  MgpuScene *scene = nullptr;
  int n_rows = 0;
  ncclComm_t comm = nullptr;
  hipStream_t comm_stream = nullptr; // every RCCL call of this communicator, in frame order
  hipStream_t rb_stream = nullptr;   // rank 0 with the read-back on: the device-to-host copies, frame after frame
  Slot slot[kMaxInFlight];
};

} // namespace

struct MgpuFrame {
  int world = 1, W = 0, H = 0, strip_h = 8, in_flight = 1;
  bool force_exchange = false; // world == 1: send the strips to ourselves through RCCL (exercises the N > 1 path on one GPU)
  // How the strips travel.  BLOCK (default): a rank's strip buffer is ONE message into a staging area on rank 0, which a
  // strided device copy per rank deals to the strips' final rows -- world - 1 receives and as many 2-D copies per frame.
  // STRIPS: one send / receive pair per strip, received at its final rows -- no staging, but 118 (1080p) or 236 (4K) pairs
  // per frame at eight ranks.  MGPU_FRAME_EXCHANGE=strips|block; measured in profiles/ (DESIGN.md 6).
  int exchange_mode = MGPU_EXCHANGE_BLOCK;
  // What carries the bytes.  RCCL (default): ncclSend / ncclRecv over xGMI.  COPY (MGPU_FRAME_TRANSPORT=copy, one process
  // driving all ranks only): device-to-device copies on rank 0's communicator stream in place of every send / receive pair --
  // same plan, same staging, same placement, same events; and because nothing in it needs one GPU per rank, several ranks may
  // then share a device: the way the N > 1 partition, staging and slot machinery is exercised for N = 2 .. 8 on a one-GPU box.
  bool transport_copy = false;
  std::vector<Member> members;  // the ranks this process drives (all of them, or one)
  unsigned long long next = 0;  // frames enqueued so far
  // exchange timing (rank 0's communicator stream): summed when a slot is waited for or reused
  double x_ms_sum = 0.0;
  unsigned long long x_frames = 0, x_ops = 0; // frames measured; receives rank 0 posts per frame
  // host time of the render calls: ONE thread enqueues launches, events and the exchange for every member of this process
  double enq_ms_sum = 0.0;
  unsigned long long enq_calls = 0;
  // SURVEY 8(d)'s frame ends with ONE read-back.  With `readback` on, every frame's copy to pinned host memory is enqueued
  // behind its exchange on a stream of its own, so it runs under the NEXT frame's kernel (frames_in_flight >= 2); the slot is
  // not rendered into again before the copy has left it.  mgpu_frame_wait_host hands the pinned buffer out.
  bool readback = false;
  bool broken = false; // a render call failed half-way: streams and slots are out of step, only destroy is allowed
  // One process driving several GPUs: the launch phase of a render call (waits, the render launch, its events -- ~40 us of host
  // time per member) is enqueued by one thread PER MEMBER, so that a call costs the slowest member's enqueue, not their sum
  // (0.33 ms per call at eight members against an ideal eighth-frame of 0.65 ms: profiles/r4_multi_ranks_on_one_gpu.txt).
  // MGPU_FRAME_ENQUEUE_THREADS=1 (opt-in).  The exchange phase stays on the caller's thread: it
  // is one RCCL group / one chain of copies on rank 0's stream.
  struct EnqueuePool *pool = nullptr;
};

namespace {

// rows of the W x H frame owned by `rank`: strips rank, rank + world, ... of strip_h rows (the last strip may be partial)
int rows_of(int H, int strip_h, int world, int rank) {
  int n = 0;
  for (int y0 = rank * strip_h; y0 < H; y0 += world * strip_h) n += (H - y0 < strip_h) ? (H - y0) : strip_h;
  return n;
}

// The exchange plan, one (offset, count) pair in floats per strip of rank `owner`, in strip order: offsets into the owner's
// local strip buffer (sender side) and into the whole frame (receiver side).  Both sides walk this one list, so the
// sends of a rank and the receives rank 0 posts for it match in number, order and size by construction.
struct Piece {
  size_t local_off, frame_off, count;
};
void plan_of(int W, int H, int strip_h, int world, int owner, std::vector<Piece> &out) {
  out.clear();
  size_t off = 0;
  for (int y0 = owner * strip_h; y0 < H; y0 += world * strip_h) {
    const size_t cnt = (size_t)3 * W * ((H - y0 < strip_h) ? (H - y0) : strip_h);
    out.push_back(Piece{off, (size_t)3 * W * y0, cnt});
    off += cnt;
  }
}

// floats in front of rank `owner`'s strip buffer in rank 0's staging area: the buffers of ranks 1 .. owner - 1 (0 .. with the
// exchange forced on one GPU), rank after rank
size_t staging_offset(int W, int H, int sh, int world, int owner, bool force) {
  size_t off = 0;
  for (int r = force ? 0 : 1; r < owner; ++r) off += (size_t)3 * rows_of(H, sh, world, r) * W;
  return off;
}

// rows the block exchange stages on rank 0: every other rank's (with the exchange forced on one GPU: its own)
size_t staging_rows(const MgpuFrame *f) {
  size_t rows = 0;
  for (int r = f->force_exchange ? 0 : 1; r < f->world; ++r) rows += (size_t)rows_of(f->H, f->strip_h, f->world, r);
  return rows;
}

int create_common(MgpuFrame *f) {
  const bool exchange = f->world > 1 || f->force_exchange;
  for (Member &m : f->members) {
    FHIP(hipSetDevice(m.device));
    m.n_rows = rows_of(f->H, f->strip_h, f->world, m.rank);
    FHIP(hipStreamCreateWithFlags(&m.comm_stream, hipStreamNonBlocking));
    for (int k = 0; k < f->in_flight; ++k) {
      Slot &s = m.slot[k];
      FHIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
      FHIP(hipEventCreateWithFlags(&s.rendered, hipEventDisableTiming));
      FHIP(hipEventCreateWithFlags(&s.exchanged, hipEventDisableTiming));
      if (m.n_rows) FHIP(hipMalloc((void **)&s.local, sizeof(float) * 3 * (size_t)m.n_rows * f->W));
      if (m.rank == 0) {
        FHIP(hipMalloc((void **)&s.frame, sizeof(float) * 3 * (size_t)f->H * f->W));
        FHIP(hipEventCreate(&s.x0));
        FHIP(hipEventCreate(&s.x1));
        if (exchange && f->exchange_mode == MGPU_EXCHANGE_BLOCK && staging_rows(f))
          FHIP(hipMalloc((void **)&s.staging, sizeof(float) * 3 * staging_rows(f) * f->W));
      }
    }
  }
  return MGPU_OK;
}

// Where the strips of rank `owner` go when its strip buffer (local rows contiguous: strip j at local rows [j * sh, ...)) is
// dealt to the frame: ONE strided 2-D copy for the full strips (strip j -> frame rows [(j * world + owner) * sh, +sh)) and one
// plain copy for a partial last strip.  All numbers in BYTES except `staging_off` / `msg_floats`; the device code below and
// mgpu_frame_block_plan (CPU tests execute it with numpy) both come through here.
struct BlockPlace {
  size_t dst_off, dst_pitch, src_pitch, width, height; // the hipMemcpy2D of the full strips (height 0: none)
  size_t tail_dst_off, tail_src_off, tail_bytes;       // the partial last strip (tail_bytes 0: none)
};
BlockPlace block_place(int rows, int owner, int world, int sh, int W) {
  const size_t strip_bytes = sizeof(float) * 3 * (size_t)sh * W;
  const int full = rows / sh, tail = rows - full * sh;
  BlockPlace p;
  p.dst_off = (size_t)owner * strip_bytes;
  p.dst_pitch = strip_bytes * world;
  p.src_pitch = strip_bytes;
  p.width = strip_bytes;
  p.height = (size_t)full;
  p.tail_dst_off = p.dst_off + (size_t)full * p.dst_pitch;
  p.tail_src_off = (size_t)full * strip_bytes;
  p.tail_bytes = sizeof(float) * 3 * (size_t)tail * W;
  return p;
}

int place_strips(float *frame, const float *local, int rows, int owner, int world, int sh, int W, hipStream_t st) {
  const BlockPlace p = block_place(rows, owner, world, sh, W);
  unsigned char *dst = reinterpret_cast<unsigned char *>(frame);
  const unsigned char *src = reinterpret_cast<const unsigned char *>(local);
  if (p.height) FHIP(hipMemcpy2DAsync(dst + p.dst_off, p.dst_pitch, src, p.src_pitch, p.width, p.height, hipMemcpyDeviceToDevice, st));
  if (p.tail_bytes) FHIP(hipMemcpyAsync(dst + p.tail_dst_off, src + p.tail_src_off, p.tail_bytes, hipMemcpyDeviceToDevice, st));
  return MGPU_OK;
}

// The stream a member's launches go to.  Slots have streams of their own so that consecutive frames' launches overlap where one
// ends and the next begins -- and, enqueued back to back, two whole-GPU persistent launches then share the CUs and
// finish TOGETHER, which is fine for throughput and useless for a read-back that wants to run under the NEXT frame's kernel
// (measured, tools/perf_copy_overlap*.py: frames complete in pairs and one copy per pair is exposed).  With the read-back on and
// one GPU, frames therefore go down ONE stream, in order.
hipStream_t render_stream(const MgpuFrame *f, const Member &m, int first_slot) {
  return (f->readback && f->world == 1) ? m.slot[0].stream : m.slot[first_slot].stream;
}

// reads a finished slot's exchange timing into the frame's sums (rank 0's member only)
void collect_timing(MgpuFrame *f, Slot &s, bool wait) {
  if (!s.x_pending) return;
  if (!wait && hipEventQuery(s.x1) != hipSuccess) { // still running: this frame's time is not booked
    (void)hipGetLastError();
    s.x_pending = false;
    return;
  }
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, s.x0, s.x1) == hipSuccess) {
    f->x_ms_sum += ms;
    f->x_frames += 1;
  } else {
    (void)hipGetLastError();
  }
  s.x_pending = false;
}

} // namespace

extern "C" {

const char *mgpu_frame_last_error(void) { return g_ferr; }

int mgpu_frame_rows(int H, int strip_h, int world, int rank) {
  if (H < 0 || strip_h <= 0 || world <= 0 || rank < 0 || rank >= world) return -1;
  return rows_of(H, strip_h, world, rank);
}

int mgpu_frame_plan(int W, int H, int strip_h, int world, int owner, size_t *local_off, size_t *frame_off, size_t *count,
                    int max_pieces) {
  if (W <= 0 || H <= 0 || strip_h <= 0 || world <= 0 || owner < 0 || owner >= world) return -1;
  std::vector<Piece> plan;
  plan_of(W, H, strip_h, world, owner, plan);
  for (int i = 0; i < (int)plan.size() && i < max_pieces; ++i) {
    if (local_off) local_off[i] = plan[i].local_off;
    if (frame_off) frame_off[i] = plan[i].frame_off;
    if (count) count[i] = plan[i].count;
  }
  return (int)plan.size();
}

int mgpu_frame_block_plan(int W, int H, int strip_h, int world, int owner, int force_exchange, size_t out[10]) {
  if (W <= 0 || H <= 0 || strip_h <= 0 || world <= 0 || owner < 0 || owner >= world || !out) return -1;
  // the staging area holds the strip buffers of ranks 1 .. world - 1 (0 .. with the exchange forced on one GPU), rank after rank
  const size_t off = staging_offset(W, H, strip_h, world, owner, force_exchange != 0);
  const int rows = rows_of(H, strip_h, world, owner);
  const BlockPlace p = block_place(rows, owner, world, strip_h, W);
  out[0] = off;
  out[1] = (size_t)3 * rows * W;
  out[2] = p.dst_off; out[3] = p.dst_pitch; out[4] = p.src_pitch; out[5] = p.width; out[6] = p.height;
  out[7] = p.tail_dst_off; out[8] = p.tail_src_off; out[9] = p.tail_bytes;
  return 0;
}

int mgpu_frame_unique_id(unsigned char id[128]) {
  if (!id) return ffail(MGPU_ERR_INVALID, "NULL argument");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId u;
  FNCCL(g_rccl.GetUniqueId(&u));
  static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id, &u, 128);
  return MGPU_OK;
}

int mgpu_frame_destroy(MgpuFrame *f) {
  if (!f) return MGPU_OK;
  delete f->pool; // (its workers are idle between render calls)
  f->pool = nullptr;
  for (Member &m : f->members) {
    (void)hipSetDevice(m.device);
    (void)hipDeviceSynchronize();
    for (Slot &s : m.slot) {
      if (s.local) (void)hipFree(s.local);
      if (s.frame) (void)hipFree(s.frame);
      if (s.staging) (void)hipFree(s.staging);
      if (s.rendered) (void)hipEventDestroy(s.rendered);
      if (s.exchanged) (void)hipEventDestroy(s.exchanged);
      if (s.x0) (void)hipEventDestroy(s.x0);
      if (s.x1) (void)hipEventDestroy(s.x1);
      if (s.stream) (void)hipStreamDestroy(s.stream);
      if (s.copied) (void)hipEventDestroy(s.copied);
      if (s.host) (void)hipHostFree(s.host);
    }
    if (m.rb_stream) (void)hipStreamDestroy(m.rb_stream);
    if (m.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m.comm);
    if (m.comm_stream) (void)hipStreamDestroy(m.comm_stream);
  }
  delete f;
  return MGPU_OK;
}

static int frame_new(int world, int W, int H, int strip_h, int frames_in_flight, MgpuFrame **out) {
  if (!out) return ffail(MGPU_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (world < 1 || W <= 0 || H <= 0 || strip_h <= 0) return ffail(MGPU_ERR_INVALID, "bad frame geometry");
  if (frames_in_flight < 1 || frames_in_flight > kMaxInFlight)
    return ffail(MGPU_ERR_INVALID, "frames_in_flight must be 1..%d", kMaxInFlight);
  MgpuFrame *f = new (std::nothrow) MgpuFrame();
  if (!f) return ffail(MGPU_ERR_OOM, "host allocation failed");
  f->world = world;
  f->W = W;
  f->H = H;
  f->strip_h = strip_h;
  f->in_flight = frames_in_flight;
  if (const char *e = getenv("MGPU_FRAME_FORCE_EXCHANGE")) f->force_exchange = world == 1 && atoi(e) != 0;
  if (const char *e = getenv("MGPU_FRAME_TRANSPORT")) {
    if (!strcmp(e, "copy")) f->transport_copy = true;
    else if (strcmp(e, "rccl") != 0) {
      delete f;
      return ffail(MGPU_ERR_INVALID, "MGPU_FRAME_TRANSPORT=%s (expected rccl|copy)", e);
    }
  }
  if (const char *e = getenv("MGPU_FRAME_EXCHANGE")) {
    if (!strcmp(e, "strips")) f->exchange_mode = MGPU_EXCHANGE_STRIPS;
    else if (!strcmp(e, "block")) f->exchange_mode = MGPU_EXCHANGE_BLOCK;
    else {
      delete f;
      return ffail(MGPU_ERR_INVALID, "MGPU_FRAME_EXCHANGE=%s (expected strips|block)", e);
    }
  }
  *out = f;
  return MGPU_OK;
}

int mgpu_frame_create(MgpuScene *const *scenes, const int *devices, int n, int W, int H, int strip_h, int frames_in_flight,
                      MgpuFrame **out) {
  if (!out) return ffail(MGPU_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!scenes || !devices || n < 1) return ffail(MGPU_ERR_INVALID, "scenes / devices must name at least one GPU");
  for (int r = 0; r < n; ++r) {
    if (!scenes[r]) return ffail(MGPU_ERR_INVALID, "scenes[%d] is NULL", r);
    if (mgpu_scene_device(scenes[r]) != devices[r])
      return ffail(MGPU_ERR_INVALID, "scenes[%d] lives on device %d, not on devices[%d] = %d", r, mgpu_scene_device(scenes[r]), r, devices[r]);
  }
  MgpuFrame *f = nullptr;
  int rc = frame_new(n, W, H, strip_h, frames_in_flight, &f);
  if (rc) return rc;
  if (!f->transport_copy) // RCCL wants one GPU per rank; the copy transport does not care
    for (int r = 0; r < n; ++r)
      for (int q = 0; q < r; ++q)
        if (devices[q] == devices[r]) {
          mgpu_frame_destroy(f);
          return ffail(MGPU_ERR_INVALID, "device %d is named twice (MGPU_FRAME_TRANSPORT=copy lets ranks share a device)", devices[r]);
        }
  f->members.resize(n);
  for (int r = 0; r < n; ++r) {
    if (!scenes[r]) {
      mgpu_frame_destroy(f);
      return ffail(MGPU_ERR_INVALID, "scenes[%d] is NULL", r);
    }
    f->members[r].rank = r;
    f->members[r].device = devices[r];
    f->members[r].scene = scenes[r];
  }
  rc = create_common(f);
  if (!rc && (n > 1 || f->force_exchange) && !f->transport_copy) {
    rc = load_rccl();
    if (!rc) {
      std::vector<ncclComm_t> comms(n);
      ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, devices);
      if (r != ncclSuccess) rc = ffail(MGPU_ERR_HIP, "ncclCommInitAll(%d devices): %s", n, g_rccl.GetErrorString(r));
      else
        for (int k = 0; k < n; ++k) f->members[k].comm = comms[k];
    }
  }
  if (rc) {
    mgpu_frame_destroy(f);
    return rc;
  }
  // (MgpuFrame::pool; opt-in until it has been measured on hardware: MGPU_FRAME_ENQUEUE_THREADS=1.)  Members that share ONE scene
  // object -- possible under MGPU_FRAME_TRANSPORT=copy, where ranks may share a device -- share its render slots, which the launch
  // phase rewrites without a lock: such a frame keeps the serial loop.
  bool scenes_distinct = true;
  for (int r = 0; r < n; ++r)
    for (int q = 0; q < r; ++q) scenes_distinct = scenes_distinct && scenes[q] != scenes[r];
  if (n >= 2 && scenes_distinct) {
    const char *e = getenv("MGPU_FRAME_ENQUEUE_THREADS");
    if (e && atoi(e) != 0) {
      try {
        f->pool = new EnqueuePool((size_t)n, [] { return g_ferr; });
      } catch (...) { // no threads to be had: the caller's thread enqueues for everybody, as without the switch
        f->pool = nullptr;
      }
    }
  }
  *out = f;
  return MGPU_OK;
}

int mgpu_frame_create_rank(MgpuScene *scene, int device, int rank, int world, const unsigned char id[128], int W, int H,
                           int strip_h, int frames_in_flight, MgpuFrame **out) {
  if (!out) return ffail(MGPU_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!scene || rank < 0 || rank >= world) return ffail(MGPU_ERR_INVALID, "bad scene / rank");
  if (mgpu_scene_device(scene) != device) return ffail(MGPU_ERR_INVALID, "the scene lives on device %d, not on device %d", mgpu_scene_device(scene), device);
  if (world > 1 && !id) return ffail(MGPU_ERR_INVALID, "a communicator id is needed for world > 1 (mgpu_frame_unique_id on one rank)");
  MgpuFrame *f = nullptr;
  int rc = frame_new(world, W, H, strip_h, frames_in_flight, &f);
  if (rc) return rc;
  if (f->transport_copy && world > 1) {
    mgpu_frame_destroy(f);
    return ffail(MGPU_ERR_UNSUPPORTED, "MGPU_FRAME_TRANSPORT=copy needs all ranks in one process (mgpu_frame_create)");
  }
  f->transport_copy = false; // world == 1 from here: RCCL to ourselves when the exchange is forced
  f->members.resize(1);
  f->members[0].rank = rank;
  f->members[0].device = device;
  f->members[0].scene = scene;
  rc = create_common(f);
  if (!rc && (world > 1 || f->force_exchange)) {
    rc = load_rccl();
    if (!rc) {
      ncclUniqueId u;
      if (id) memcpy(&u, id, 128);
      else if (g_rccl.GetUniqueId(&u) != ncclSuccess) rc = ffail(MGPU_ERR_HIP, "ncclGetUniqueId failed");
      if (!rc) {
        hipError_t e = hipSetDevice(device);
        ncclResult_t r = e == hipSuccess ? g_rccl.CommInitRank(&f->members[0].comm, world, u, rank) : ncclUnhandledCudaError;
        if (r != ncclSuccess) rc = ffail(MGPU_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(r));
      }
    }
  }
  if (rc) {
    mgpu_frame_destroy(f);
    return rc;
  }
  *out = f;
  return MGPU_OK;
}

// n frames (n = 1: mgpu_frame_render): every member renders its strips of all n frames with ONE launch on the first slot's
// stream; then, on the communicator streams and frame by frame, the strips travel to rank 0 (one grouped exchange step per
// frame, see MgpuFrame::exchange_mode); rank 0's own strips are placed by one strided device copy.
// A failure after the first enqueue leaves streams, events and the frame counter out of step: the frame is marked broken and
// every later call except mgpu_frame_destroy is refused (an RCCL group opened here is closed before returning).
static int render_frames_enqueue(MgpuFrame *f, const double cam[12], int maxPathLength, int passes, const float plane[4], int rng_mode,
                                 uint64_t seed, uint32_t pass_base, int n, int *slots_out) {
  int ks[kMaxInFlight];
  for (int i = 0; i < n; ++i) ks[i] = (int)((f->next + (unsigned long long)i) % (unsigned long long)f->in_flight);
  const int W = f->W, H = f->H, sh = f->strip_h, world = f->world;
  const bool exchange = world > 1 || f->force_exchange;
  const bool block = f->exchange_mode == MGPU_EXCHANGE_BLOCK;
  // the launch phase of one member (its own device, streams and events: members do not touch each other's here)
  auto launch_member = [&](size_t mi) -> int {
    Member &m = f->members[mi];
    FHIP(hipSetDevice(m.device));
    hipStream_t rs = render_stream(f, m, ks[0]); // the launch and the copies of the whole batch
    // the slots' previous frames must have left their buffers: their exchange is the last thing that touched them
    for (int i = 0; i < n; ++i) {
      FHIP(hipStreamWaitEvent(rs, m.slot[ks[i]].exchanged, 0));
      if (m.slot[ks[i]].copy_pending) FHIP(hipStreamWaitEvent(rs, m.slot[ks[i]].copied, 0)); // ... and its read-back
      if (m.rank == 0) collect_timing(f, m.slot[ks[i]], false);
    }
    // one rank and no forced exchange: its "strips" are the whole frame in frame order -- rendered straight into the frame
    // buffer (the strided device copy that deals strips to their rows cost 0.07 ms of a 5.5 ms frame for moving nothing)
    const bool direct = world == 1 && !f->force_exchange && m.rank == 0;
    if (m.n_rows) {
      float *images[kMaxInFlight];
      for (int i = 0; i < n; ++i) images[i] = direct ? m.slot[ks[i]].frame : m.slot[ks[i]].local;
      int rc = mgpu_render_frames_device(m.scene, cam, W, H, 0, W, m.rank * sh, sh, sh * world, m.n_rows, maxPathLength, passes, plane,
                                         rng_mode, nullptr, seed, pass_base, n, images, nullptr, rs, nullptr);
      if (rc) return ffail(rc, "rank %d: %s", m.rank, mgpu_last_error());
    }
    for (int i = 0; i < n; ++i) {
      Slot &s = m.slot[ks[i]];
      if (m.rank == 0 && m.n_rows && !f->force_exchange && !direct) { // own strips to their final rows
        int rc = place_strips(s.frame, s.local, m.n_rows, 0, world, sh, W, rs);
        if (rc) return rc;
      }
      FHIP(hipEventRecord(s.rendered, rs));
    }
    if (exchange) FHIP(hipStreamWaitEvent(m.comm_stream, m.slot[ks[n - 1]].rendered, 0));
    return MGPU_OK;
  };
  if (f->pool) {
    const std::function<int(size_t)> job = launch_member;
    const int rc = f->pool->run(job);
    if (rc) return rc;
  } else {
    for (size_t mi = 0; mi < f->members.size(); ++mi) {
      const int rc = launch_member(mi);
      if (rc) return rc;
    }
  }
  Member *root = nullptr; // the member that holds rank 0 (copy transport: the one that moves everybody's bytes)
  for (Member &m : f->members)
    if (m.rank == 0) root = &m;
  if (exchange && f->transport_copy) {
    if (!root) return ffail(MGPU_ERR_INVALID, "copy transport without rank 0 in this process");
    FHIP(hipSetDevice(root->device));
    for (Member &m : f->members) // rank 0's communicator stream moves the bytes: it waits for every rank's launch
      if (&m != root) FHIP(hipStreamWaitEvent(root->comm_stream, m.slot[ks[n - 1]].rendered, 0));
  }
  for (int i = 0; i < n; ++i) {
    const int k = ks[i];
    if (exchange && f->transport_copy) {
      // the exchange step with device-to-device copies in place of the send / receive pairs: same plan, staging and placement
      FHIP(hipSetDevice(root->device));
      FHIP(hipEventRecord(root->slot[k].x0, root->comm_stream));
      unsigned long long ops = 0;
      std::vector<Piece> plan;
      for (Member &m : f->members) {
        if (m.rank == 0 && !f->force_exchange) continue;
        Slot &src = m.slot[k];
        if (block) {
          const size_t cnt = (size_t)3 * m.n_rows * W;
          if (cnt)
            FHIP(hipMemcpyAsync(root->slot[k].staging + staging_offset(W, H, sh, world, m.rank, f->force_exchange), src.local, sizeof(float) * cnt,
                                hipMemcpyDefault, root->comm_stream));
          ops += cnt ? 1 : 0;
          int rc = place_strips(root->slot[k].frame, root->slot[k].staging + staging_offset(W, H, sh, world, m.rank, f->force_exchange), m.n_rows,
                                m.rank, world, sh, W, root->comm_stream);
          if (rc) return rc;
        } else {
          plan_of(W, H, sh, world, m.rank, plan);
          for (const Piece &p : plan) {
            FHIP(hipMemcpyAsync(root->slot[k].frame + p.frame_off, src.local + p.local_off, sizeof(float) * p.count, hipMemcpyDefault, root->comm_stream));
            ops += 1;
          }
        }
      }
      f->x_ops = ops;
      FHIP(hipEventRecord(root->slot[k].x1, root->comm_stream));
      root->slot[k].x_pending = true;
      FHIP(hipEventRecord(root->slot[k].exchanged, root->comm_stream));
      for (Member &m : f->members) // a rank's strip buffer is free when rank 0 has taken its bytes
        if (&m != root) {
          FHIP(hipSetDevice(m.device));
          FHIP(hipStreamWaitEvent(m.comm_stream, root->slot[k].exchanged, 0));
          FHIP(hipEventRecord(m.slot[k].exchanged, m.comm_stream));
        }
      if (slots_out) slots_out[i] = k;
      continue;
    }
    if (exchange) {
      for (Member &m : f->members)
        if (m.rank == 0) {
          FHIP(hipSetDevice(m.device));
          FHIP(hipEventRecord(m.slot[k].x0, m.comm_stream));
        }
      FNCCL(g_rccl.GroupStart());
      // inside the group an error must not return before the group is closed
      int grc = MGPU_OK;
      unsigned long long ops = 0;
      auto nccl_ok = [&](ncclResult_t r, const char *what) {
        if (r != ncclSuccess && grc == MGPU_OK) grc = ffail(MGPU_ERR_HIP, "%s failed: %s", what, g_rccl.GetErrorString(r));
        return r == ncclSuccess;
      };
      for (Member &m : f->members) {
        Slot &s = m.slot[k];
        std::vector<Piece> plan;
        if (m.rank != 0 || f->force_exchange) { // sends: this rank's strips -- as one block, or strip by strip in strip order
          if (block) {
            if (m.n_rows) nccl_ok(g_rccl.Send(s.local, (size_t)3 * m.n_rows * W, ncclFloat, 0, m.comm, m.comm_stream), "ncclSend");
          } else {
            plan_of(W, H, sh, world, m.rank, plan);
            for (const Piece &p : plan)
              if (!nccl_ok(g_rccl.Send(s.local + p.local_off, p.count, ncclFloat, 0, m.comm, m.comm_stream), "ncclSend")) break;
          }
        }
        if (m.rank == 0) { // receives: every other rank's strips (its own too when the exchange is forced)
          for (int r = f->force_exchange ? 0 : 1; r < world && grc == MGPU_OK; ++r) {
            if (block) { // rank r's whole strip buffer, behind the previous rank's in the staging area
              const size_t cnt = (size_t)3 * rows_of(H, sh, world, r) * W;
              if (cnt) nccl_ok(g_rccl.Recv(s.staging + staging_offset(W, H, sh, world, r, f->force_exchange), cnt, ncclFloat, r, m.comm, m.comm_stream), "ncclRecv");
              ops += cnt ? 1 : 0;
            } else { // every strip at its final rows
              plan_of(W, H, sh, world, r, plan);
              for (const Piece &p : plan) {
                if (!nccl_ok(g_rccl.Recv(s.frame + p.frame_off, p.count, ncclFloat, r, m.comm, m.comm_stream), "ncclRecv")) break;
                ops += 1;
              }
            }
          }
        }
      }
      const ncclResult_t ge = g_rccl.GroupEnd();
      if (grc != MGPU_OK) return grc;
      FNCCL(ge);
      f->x_ops = ops;
      if (block) // deal the staged buffers to their rows: one strided copy per rank, behind the receives on the same stream
        for (Member &m : f->members)
          if (m.rank == 0) {
            FHIP(hipSetDevice(m.device));
            for (int r = f->force_exchange ? 0 : 1; r < world; ++r) {
              const int rows = rows_of(H, sh, world, r);
              int rc = place_strips(m.slot[k].frame, m.slot[k].staging + staging_offset(W, H, sh, world, r, f->force_exchange), rows, r, world, sh, W,
                                    m.comm_stream);
              if (rc) return rc;
            }
          }
    }
    for (Member &m : f->members) {
      FHIP(hipSetDevice(m.device));
      if (exchange && m.rank == 0) {
        FHIP(hipEventRecord(m.slot[k].x1, m.comm_stream));
        m.slot[k].x_pending = true;
      }
      FHIP(hipEventRecord(m.slot[k].exchanged, exchange ? m.comm_stream : render_stream(f, m, ks[0])));
    }
    if (slots_out) slots_out[i] = k;
  }
  if (f->readback && root) {
    if (f->world == 1) {
      for (int i = 0; i < n; ++i) root->slot[ks[i]].copy_wanted = true; // issued by mgpu_frame_wait_host, see there
    } else {
      // Several GPUs: rank 0 renders an N-th of a frame while a whole frame crosses PCIe, and its host thread is busy enqueueing for
      // everybody -- the copies are enqueued HERE, each behind its frame's exchange on the read-back stream, so that they run as the
      // frames arrive and only the last one is left when the caller comes to take them (the stream-side wait costs rank 0's GPU
      // ~0.1 ms per frame: tools/perf_copy_overlap3.py; a batch of eight taken lazily at the end of a short run is 3.7 ms of idle GPUs)
      FHIP(hipSetDevice(root->device));
      for (int i = 0; i < n; ++i) {
        Slot &s = root->slot[ks[i]];
        FHIP(hipStreamWaitEvent(root->rb_stream, s.exchanged, 0));
        FHIP(hipMemcpyAsync(s.host, s.frame, sizeof(float) * 3 * (size_t)f->W * f->H, hipMemcpyDeviceToHost, root->rb_stream));
        FHIP(hipEventRecord(s.copied, root->rb_stream));
        s.copy_wanted = false;
        s.copy_pending = true;
      }
    }
  }
  f->next += (unsigned long long)n;
  return MGPU_OK;
}

static int render_frames(MgpuFrame *f, const double cam[12], int maxPathLength, int passes, const float plane[4], int rng_mode,
                         uint64_t seed, uint32_t pass_base, int n, int *slots_out) {
  if (!f || !cam) return ffail(MGPU_ERR_INVALID, "NULL argument");
  if (f->broken) return ffail(MGPU_ERR_INVALID, "this frame object failed in an earlier render call and can only be destroyed");
  if (n < 1 || n > f->in_flight) return ffail(MGPU_ERR_INVALID, "n_frames must be 1..frames_in_flight (%d)", f->in_flight);
  // what can be refused before anything is enqueued is refused here and leaves the frame object usable
  if (maxPathLength < 1 || passes < 1) return ffail(MGPU_ERR_INVALID, "maxPathLength and passes must be >= 1");
  if (rng_mode != MGPU_RNG_HASH)
    return ffail(MGPU_ERR_UNSUPPORTED, "multi-GPU frames are seeded per (pixel, pass) (MGPU_RNG_HASH): the image must not depend on the GPU count");
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = render_frames_enqueue(f, cam, maxPathLength, passes, plane, rng_mode, seed, pass_base, n, slots_out);
  f->enq_ms_sum += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  f->enq_calls += 1;
  if (rc) f->broken = true;
  return rc;
}

int mgpu_frame_render(MgpuFrame *f, const double cam[12], int maxPathLength, int passes, const float plane[4], int rng_mode,
                      uint64_t seed, uint32_t pass_base, int *slot_out) {
  return render_frames(f, cam, maxPathLength, passes, plane, rng_mode, seed, pass_base, 1, slot_out);
}

int mgpu_frame_render_batch(MgpuFrame *f, const double cam[12], int maxPathLength, int passes, const float plane[4], int rng_mode,
                            uint64_t seed, uint32_t pass_base, int n_frames, int *slots_out) {
  return render_frames(f, cam, maxPathLength, passes, plane, rng_mode, seed, pass_base, n_frames, slots_out);
}

int mgpu_frame_wait(MgpuFrame *f, int slot, float *host_image, float **device_image) {
  if (!f || slot < 0 || slot >= f->in_flight) return ffail(MGPU_ERR_INVALID, "bad frame / slot");
  float *dev = nullptr;
  for (Member &m : f->members) {
    FHIP(hipSetDevice(m.device));
    FHIP(hipEventSynchronize(m.slot[slot].exchanged));
    if (m.rank == 0) {
      dev = m.slot[slot].frame;
      collect_timing(f, m.slot[slot], true);
    }
  }
  if (device_image) *device_image = dev;
  if (host_image) {
    if (!dev) return ffail(MGPU_ERR_INVALID, "this process does not hold rank 0: the frame lives elsewhere");
    FHIP(hipMemcpy(host_image, dev, sizeof(float) * 3 * (size_t)f->W * f->H, hipMemcpyDeviceToHost));
  }
  return MGPU_OK;
}

int mgpu_frame_set_readback(MgpuFrame *f, int on) {
  if (!f) return ffail(MGPU_ERR_INVALID, "NULL argument");
  if (f->broken) return ffail(MGPU_ERR_INVALID, "this frame object failed in an earlier render call and can only be destroyed");
  for (Member &m : f->members) {
    if (m.rank != 0) continue;
    FHIP(hipSetDevice(m.device));
    if (on && !m.rb_stream) FHIP(hipStreamCreateWithFlags(&m.rb_stream, hipStreamNonBlocking));
    for (int k = 0; k < f->in_flight; ++k) {
      Slot &s = m.slot[k];
      if (s.copy_pending) { // a switch between frames: nothing of the old mode stays in flight
        FHIP(hipEventSynchronize(s.copied));
        s.copy_pending = false;
      }
      s.copy_wanted = false;
      if (s.exchanged) FHIP(hipEventSynchronize(s.exchanged)); // the render stream changes with the mode (render_stream)
      if (on && !s.host) {
        FHIP(hipHostMalloc((void **)&s.host, sizeof(float) * 3 * (size_t)f->W * f->H, hipHostMallocDefault));
        FHIP(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
      }
    }
  }
  f->readback = on != 0;
  return MGPU_OK;
}

int mgpu_frame_wait_host(MgpuFrame *f, int slot, const float **host_image) {
  if (!f || slot < 0 || slot >= f->in_flight || !host_image) return ffail(MGPU_ERR_INVALID, "bad frame / slot / NULL argument");
  *host_image = nullptr;
  for (Member &m : f->members) // every other member first: its strips have left when this returns, wherever rank 0 stands in the list
    if (m.rank != 0) {
      FHIP(hipSetDevice(m.device));
      FHIP(hipEventSynchronize(m.slot[slot].exchanged));
    }
  for (Member &m : f->members) {
    if (m.rank != 0) continue;
    FHIP(hipSetDevice(m.device));
    Slot &s = m.slot[slot];
    if (!(s.copy_wanted || s.copy_pending) || !s.host)
      return ffail(MGPU_ERR_INVALID, "slot %d has no read-back in flight (mgpu_frame_set_readback before the render call)", slot);
    if (s.copy_wanted) {
      // The copy is enqueued by the HOST once it has seen the frame complete, not by a stream-side wait: a copy that waits on
      // the GPU for the frame's event costs 0.12 ms of a 5.5 ms frame more (the runtime then orders it through the compute
      // queue), one enqueued after the fact goes straight to the copy engine and disappears under the next frame's kernel
      // (5.54 ms per C2 frame with and without it; tools/perf_copy_overlap3.py).
      FHIP(hipEventSynchronize(s.exchanged));
      FHIP(hipMemcpyAsync(s.host, s.frame, sizeof(float) * 3 * (size_t)f->W * f->H, hipMemcpyDeviceToHost, m.rb_stream));
      FHIP(hipEventRecord(s.copied, m.rb_stream));
      s.copy_wanted = false;
      s.copy_pending = true;
    }
    FHIP(hipEventSynchronize(s.copied));
    collect_timing(f, s, true);
    *host_image = s.host;
    return MGPU_OK;
  }
  return MGPU_OK; // a process without rank 0: its strips have left (exchanged), the frame lives elsewhere (*host_image stays NULL)
}

int mgpu_frame_stats(MgpuFrame *f, MgpuFrameStats *out, int reset) {
  if (!f || !out) return ffail(MGPU_ERR_INVALID, "NULL argument");
  memset(out, 0, sizeof(*out));
  out->world = f->world;
  out->members = (int)f->members.size();
  out->exchange_mode = f->exchange_mode + (f->transport_copy ? 2 : 0);
  out->frames = f->next;
  for (Member &m : f->members) {
    if (m.comm && out->rccl_ranks == 0) { // what the communicator itself says about its size
      int cnt = 0;
      FNCCL(g_rccl.CommCount(m.comm, &cnt));
      out->rccl_ranks = cnt;
    }
    if (m.rank == 0) {
      FHIP(hipSetDevice(m.device));
      for (int k = 0; k < f->in_flight; ++k)
        if (m.slot[k].x_pending) {
          FHIP(hipEventSynchronize(m.slot[k].x1));
          collect_timing(f, m.slot[k], true);
        }
    }
  }
  out->exchange_frames = f->x_frames;
  out->exchange_ms = f->x_ms_sum;
  out->exchange_ops_per_frame = f->x_ops;
  out->enqueue_calls = f->enq_calls;
  out->enqueue_ms = f->enq_ms_sum;
  if (reset) {
    f->x_frames = 0;
    f->x_ms_sum = 0.0;
    f->enq_calls = 0;
    f->enq_ms_sum = 0.0;
  }
  return MGPU_OK;
}

int mgpu_frame_done_event_wait(MgpuFrame *f, int slot, void *stream) {
  // makes `stream` (a hipStream_t of rank 0's device) wait for the slot's frame without blocking the host
  if (!f || slot < 0 || slot >= f->in_flight) return ffail(MGPU_ERR_INVALID, "bad frame / slot");
  for (Member &m : f->members)
    if (m.rank == 0) {
      FHIP(hipSetDevice(m.device));
      FHIP(hipStreamWaitEvent((hipStream_t)stream, m.slot[slot].exchanged, 0));
      return MGPU_OK;
    }
  return ffail(MGPU_ERR_INVALID, "this process does not hold rank 0");
}

} // extern "C"
