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
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mgpu.h"

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

constexpr int kMaxInFlight = 8;

// what one device keeps for one frame in flight
struct Slot {
  hipStream_t stream = nullptr; // render stream of this slot
  float *local = nullptr;       // this rank's strips, local rows contiguous (n_rows x W x 3)
  float *frame = nullptr;       // rank 0 only: the whole frame (H x W x 3)
  hipEvent_t rendered = nullptr, exchanged = nullptr;
};

struct Member { // one GPU of this process
  int rank = 0, device = 0;
  MgpuScene *scene = nullptr;
  int n_rows = 0;
  ncclComm_t comm = nullptr;
  hipStream_t comm_stream = nullptr; // every RCCL call of this communicator, in frame order
  Slot slot[kMaxInFlight];
};

} // namespace

struct MgpuFrame {
  int world = 1, W = 0, H = 0, strip_h = 8, in_flight = 1;
  bool force_exchange = false; // world == 1: send the strips to ourselves through RCCL (exercises the N > 1 path on one GPU)
  std::vector<Member> members;  // the ranks this process drives (all of them, or one)
  unsigned long long next = 0;  // frames enqueued so far
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

int create_common(MgpuFrame *f) {
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
      if (m.rank == 0) FHIP(hipMalloc((void **)&s.frame, sizeof(float) * 3 * (size_t)f->H * f->W));
    }
  }
  return MGPU_OK;
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
  for (Member &m : f->members) {
    (void)hipSetDevice(m.device);
    (void)hipDeviceSynchronize();
    for (Slot &s : m.slot) {
      if (s.local) (void)hipFree(s.local);
      if (s.frame) (void)hipFree(s.frame);
      if (s.rendered) (void)hipEventDestroy(s.rendered);
      if (s.exchanged) (void)hipEventDestroy(s.exchanged);
      if (s.stream) (void)hipStreamDestroy(s.stream);
    }
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
  *out = f;
  return MGPU_OK;
}

int mgpu_frame_create(MgpuScene *const *scenes, const int *devices, int n, int W, int H, int strip_h, int frames_in_flight,
                      MgpuFrame **out) {
  if (!scenes || !devices || n < 1) return ffail(MGPU_ERR_INVALID, "scenes / devices must name at least one GPU");
  MgpuFrame *f = nullptr;
  int rc = frame_new(n, W, H, strip_h, frames_in_flight, &f);
  if (rc) return rc;
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
  if (!rc && (n > 1 || f->force_exchange)) {
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
  *out = f;
  return MGPU_OK;
}

int mgpu_frame_create_rank(MgpuScene *scene, int device, int rank, int world, const unsigned char id[128], int W, int H,
                           int strip_h, int frames_in_flight, MgpuFrame **out) {
  if (!scene || rank < 0 || rank >= world) return ffail(MGPU_ERR_INVALID, "bad scene / rank");
  if (world > 1 && !id) return ffail(MGPU_ERR_INVALID, "a communicator id is needed for world > 1 (mgpu_frame_unique_id on one rank)");
  MgpuFrame *f = nullptr;
  int rc = frame_new(world, W, H, strip_h, frames_in_flight, &f);
  if (rc) return rc;
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
// stream; then, on the communicator streams and frame by frame, the strips travel to rank 0's frame (grouped send / recv,
// one pair per strip, received at the strip's final rows); rank 0's own strips are placed by one strided device copy.
static int render_frames(MgpuFrame *f, const double cam[12], int maxPathLength, int passes, const float plane[4], int rng_mode,
                         uint64_t seed, uint32_t pass_base, int n, int *slots_out) {
  if (!f || !cam) return ffail(MGPU_ERR_INVALID, "NULL argument");
  if (n < 1 || n > f->in_flight) return ffail(MGPU_ERR_INVALID, "n_frames must be 1..frames_in_flight (%d)", f->in_flight);
  int ks[kMaxInFlight];
  for (int i = 0; i < n; ++i) ks[i] = (int)((f->next + (unsigned long long)i) % (unsigned long long)f->in_flight);
  const int W = f->W, H = f->H, sh = f->strip_h, world = f->world;
  const size_t strip_floats = (size_t)3 * sh * W;
  const bool exchange = world > 1 || f->force_exchange;
  for (Member &m : f->members) {
    FHIP(hipSetDevice(m.device));
    hipStream_t rs = m.slot[ks[0]].stream; // the launch and the copies of the whole batch
    // the slots' previous frames must have left their buffers: their exchange is the last thing that touched them
    for (int i = 0; i < n; ++i) FHIP(hipStreamWaitEvent(rs, m.slot[ks[i]].exchanged, 0));
    if (m.n_rows) {
      float *images[kMaxInFlight];
      for (int i = 0; i < n; ++i) images[i] = m.slot[ks[i]].local;
      int rc = mgpu_render_frames_device(m.scene, cam, W, H, 0, W, m.rank * sh, sh, sh * world, m.n_rows, maxPathLength, passes, plane,
                                         rng_mode, nullptr, seed, pass_base, n, images, nullptr, rs, nullptr);
      if (rc) return ffail(rc, "rank %d: %s", m.rank, mgpu_last_error());
    }
    for (int i = 0; i < n; ++i) {
      Slot &s = m.slot[ks[i]];
      if (m.rank == 0 && m.n_rows && !f->force_exchange) {
        // own strips to their final rows: local strip j -> frame rows [j * world * sh, +sh); the last one may be partial
        const int full = m.n_rows / sh, tail = m.n_rows - full * sh;
        if (full)
          FHIP(hipMemcpy2DAsync(s.frame, sizeof(float) * strip_floats * world, s.local, sizeof(float) * strip_floats,
                                sizeof(float) * strip_floats, (size_t)full, hipMemcpyDeviceToDevice, rs));
        if (tail)
          FHIP(hipMemcpyAsync(s.frame + (size_t)full * world * strip_floats, s.local + (size_t)full * strip_floats,
                              sizeof(float) * 3 * (size_t)tail * W, hipMemcpyDeviceToDevice, rs));
      }
      FHIP(hipEventRecord(s.rendered, rs));
    }
    if (exchange) FHIP(hipStreamWaitEvent(m.comm_stream, m.slot[ks[n - 1]].rendered, 0));
  }
  for (int i = 0; i < n; ++i) {
    const int k = ks[i];
    if (exchange) {
      FNCCL(g_rccl.GroupStart());
      for (Member &m : f->members) {
        Slot &s = m.slot[k];
        std::vector<Piece> plan;
        if (m.rank != 0 || f->force_exchange) { // sends: this rank's strips, in strip order
          plan_of(W, H, sh, world, m.rank, plan);
          for (const Piece &p : plan) FNCCL(g_rccl.Send(s.local + p.local_off, p.count, ncclFloat, 0, m.comm, m.comm_stream));
        }
        if (m.rank == 0) { // receives: every other rank's strips (its own too when the exchange is forced), at their final rows
          for (int r = f->force_exchange ? 0 : 1; r < world; ++r) {
            plan_of(W, H, sh, world, r, plan);
            for (const Piece &p : plan) FNCCL(g_rccl.Recv(s.frame + p.frame_off, p.count, ncclFloat, r, m.comm, m.comm_stream));
          }
        }
      }
      FNCCL(g_rccl.GroupEnd());
    }
    for (Member &m : f->members) {
      FHIP(hipSetDevice(m.device));
      FHIP(hipEventRecord(m.slot[k].exchanged, exchange ? m.comm_stream : m.slot[ks[0]].stream));
    }
    if (slots_out) slots_out[i] = k;
  }
  f->next += (unsigned long long)n;
  return MGPU_OK;
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
    if (m.rank == 0) dev = m.slot[slot].frame;
  }
  if (device_image) *device_image = dev;
  if (host_image) {
    if (!dev) return ffail(MGPU_ERR_INVALID, "this process does not hold rank 0: the frame lives elsewhere");
    FHIP(hipMemcpy(host_image, dev, sizeof(float) * 3 * (size_t)f->W * f->H, hipMemcpyDeviceToHost));
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
