/*
 * include/mgpu.h -- C ABI of the MI355X render hot path (libmallie_mgpu.so).
 *
 * This is the drop-in boundary underneath the Mallie-compatible C++ facade (the headers under include/mallie/): plain pointers and
 * sizes, no C++ or torch types.  Every entry point names the reference interface it replaces (paths relative to the
 * lighttransport/mallie tree).  All functions return MGPU_OK (0) on success and a negative code on failure; nothing
 * throws.  The caller owns every buffer it passes.  One host thread per scene handle at a time.
 *
 * The library contains NO CPU fallback: if no HIP device is usable every call fails with MGPU_ERR_NO_DEVICE.
 */
#ifndef MGPU_H_
#define MGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGPU_ABI_VERSION 1

enum {
  MGPU_OK = 0,
  MGPU_ERR_INVALID = -1,     /* bad argument                                                        */
  MGPU_ERR_NO_DEVICE = -2,   /* no usable HIP device / device index out of range                     */
  MGPU_ERR_OOM = -3,         /* host or device allocation failed                                     */
  MGPU_ERR_HIP = -4,         /* a HIP runtime call failed (see mgpu_last_error)                      */
  MGPU_ERR_STACK = -5,       /* traversal needed more stack than the reference's 512 entries         */
  MGPU_ERR_UNSUPPORTED = -6  /* e.g. MGPU_RNG_STREAM requested from the device renderer              */
};

/* RNG start-state source for mgpu_render* (SURVEY.md H1). The generator itself is the reference's xorshift128
 * (render.cc:137-168); only where a path's 128-bit start state comes from differs. */
enum {
  MGPU_RNG_STREAM = 0, /* the reference's serial thread stream (OMP_NUM_THREADS=1): mgpu_render_stream; the other entry
                        * points return MGPU_ERR_UNSUPPORTED for it                                                   */
  MGPU_RNG_TABLE = 1,  /* rng_states[((pass*H + y)*W + x)*4 .. +4]: start state of every (pass, pixel)              */
  MGPU_RNG_HASH = 2    /* start state = hash(seed, pass_base + pass, y*W + x)  (mgpu_hash_state)                    */
};

/* 64-byte BVH node, byte-identical to the reference's BVHNode (bvh_accel.h:10-30). */
typedef struct {
  double bmin[3];
  double bmax[3];
  int32_t flag;     /* 1 = leaf, 0 = branch */
  int32_t axis;     /* split axis of a branch */
  uint32_t data[2]; /* leaf: {count, first index into indices}; branch: {child0, child1} */
} MgpuNode;

/* 88-byte ray, byte-identical to the reference's Ray (common.h:78-83); only org and dir are read. */
typedef struct {
  double org[3];
  double dir[3];
  double invDir[3];
  int32_t dirSign[3];
  int32_t pad_;
} MgpuRay;

/* 184-byte hit record, byte-identical to the reference's Intersection (intersection.h:6-24). */
typedef struct {
  double t, u, v;
  uint32_t faceID, materialID;
  uint32_t f0, f1, f2;
  uint32_t pad_;
  double position[3];
  double geometricNormal[3];
  double normal[3];
  double tangent[3];
  double binormal[3];
  double texcoord[2];
} MgpuIntersection;

/* Work counters of one call (events, not bytes). "Real" rays are Scene::Trace calls made up to and including a path's
 * first miss; the reference's post-miss continuation rays are finished analytically on the device and
 * are reported only through trace_calls. */
typedef struct {
  uint64_t trace_calls;  /* reference-equivalent Scene::Trace() calls                */
  uint64_t real_rays;    /* BVH traversals actually performed                         */
  uint64_t nodes;        /* BVH nodes popped and box-tested                           */
  uint64_t tris;         /* ray-triangle tests                                        */
  uint64_t paths;        /* eye paths (= pixels * passes)                             */
  uint64_t stack_overflow; /* != 0: some ray exceeded 512 stack entries (result invalid) */
  double kernel_ms;      /* device time of the dominant kernel, HIP events on its stream */
  double total_ms;       /* whole call incl. copies, host wall clock                   */
} MgpuStats;

typedef struct MgpuScene MgpuScene;

/* -- library ------------------------------------------------------------------------------------------------------ */
int mgpu_abi_version(void);
int mgpu_device_count(void);                 /* number of usable HIP devices, 0 if none */
const char *mgpu_last_error(void);           /* thread-local text of the last failure   */
const char *mgpu_status_string(int status);

/* -- scene: replaces the data Scene::Init leaves behind (scene.cc:66-251: Mesh arrays + BVHAccel nodes_/indices_) --- */
/* verts 3*nv doubles; faces 3*nf; matIDs nf or NULL (then hits report 0xFFFFFFFF, bvh_accel.cc:687-691);
 * fv_normals 9*nf or NULL (mesh.h:13); fv_uvs 6*nf or NULL; nodes/indices as BVHAccel::GetNodes()/GetIndices()
 * (bvh_accel.h:74-75); mat_diffuse 3*nm doubles = Scene::materials_[i].diffuse (nm may be 0: default 0.5 grey,
 * scene.h:58-65).  Everything is copied to `device` and re-laid-out there; the host arrays may be freed afterwards. */
int mgpu_scene_create(const double *verts, size_t nv, const uint32_t *faces, size_t nf, const uint32_t *matIDs,
                      const double *fv_normals, const double *fv_uvs, const MgpuNode *nodes, size_t nn,
                      const uint32_t *indices, const double *mat_diffuse, size_t nm, int device, MgpuScene **out);
int mgpu_scene_destroy(MgpuScene *scene);

/* Arithmetic of the render entry points of this scene (mgpu_render, mgpu_render_strips_device, mgpu_render_frames_device,
 * mgpu_frame_*): MGPU_PRECISION_FP64 (default) = the reference's double arithmetic, bit-identical to the reference;
 * MGPU_PRECISION_FP32 = the FAST MODE: the same algorithm, same random stream, same visiting order in
 * float on a float copy of the scene (32-byte nodes with outward-rounded boxes, 48-byte triangles; built on first use, +56 %
 * of the node / triangle memory).  Not bit-exact: a path follows the reference's path until a hit / miss decision falls
 * differently (rays within ~1e-6 of a silhouette) and is another sample of the same integrand from there; measured distance
 * to the fp64 frame and speed: DESIGN.md 5.  mgpu_trace*, mgpu_render_aov, mgpu_render_stream, mgpu_render_step and
 * mgpu_render_panoramic* always compute in double.  Like mgpu_scene_destroy, not to be called while another thread is
 * inside a render call of the same scene; launches already enqueued keep the arithmetic they were enqueued with. */
#define MGPU_PRECISION_FP64 0
#define MGPU_PRECISION_FP32 1
int mgpu_scene_set_precision(MgpuScene *scene, int precision);
/* Scene::BoundingBox (scene.cc:317-333). */
int mgpu_scene_bbox(const MgpuScene *scene, double bmin[3], double bmax[3]);
/* Bytes resident on the device for this scene. */
size_t mgpu_scene_device_bytes(const MgpuScene *scene);
/* The HIP device the scene was created on (-1 for NULL). */
int mgpu_scene_device(const MgpuScene *scene);

/* -- batched Scene::Trace (scene.cc:253-315 -> BVHAccel::Traverse, bvh_accel.cc:773-844) ------------------------------ */
/* For each of n rays: out[i] = the Intersection Traverse would fill, hit[i] = its bool result. On a miss out[i] has
 * t = DBL_MAX, u = v = 0, faceID = 0xFFFFFFFF and all other fields zero. stats may be NULL. */
int mgpu_trace(MgpuScene *scene, const MgpuRay *rays, size_t n, MgpuIntersection *out, uint8_t *hit, MgpuStats *stats);
/* Calls of mgpu_trace with 2..64 rays and stats == NULL (and one-ray calls when the resident server below is switched off) --
 * small requests from many host threads, as the reference's OpenMP loops make them (scene.cc:253-315, render.cc:403) -- go
 * through a submission queue: the calls that are
 * inside mgpu_trace at the same time are served by ONE launch (one of the callers packs everybody's rays into host memory the
 * device maps, the traversal writes the records straight back, no copy engine in the path), a caller that is alone waits for
 * nobody.  Same records as a launch per call.  MGPU_TRACE_QUEUE=0 (read when the scene is created) switches it off.
 * mgpu_trace_queue_stats: combined launches so far and the calls they served. */
int mgpu_trace_queue_stats(MgpuScene *scene, uint64_t *launches, uint64_t *calls);
/* Calls of mgpu_trace with exactly ONE ray and stats == NULL -- Scene::Trace as the reference calls it -- launch nothing while a
 * RESIDENT SERVER is alive: the caller writes its ray and a request number into a mailbox slot in host memory the device maps, a
 * resident wave (k_trace_server, 16 waves, the same traversal as the batched kernel) writes the Intersection back beside its
 * acknowledgement.  The first call (and the first after a pause) launches the server; it leaves by itself after
 * MGPU_TRACE_SERVER_IDLE_US (default 1000) without requests and at the latest MGPU_TRACE_SERVER_LIFE_US (default 100000) after
 * its launch, so a hipDeviceSynchronize / hipFree elsewhere in the process waits at most that long; the render entry points
 * retire it before they launch.  Same records as the batched path.  MGPU_TRACE_SERVER=0 (read when the scene is created) sends
 * one-ray calls through the submission queue instead.
 * mgpu_trace_server_stats: launches of the server so far, calls it served, whether a launch is alive right now, and (device_us,
 * may be NULL) the mean device time of a call from the poll that found it to its acknowledgement.
 * mgpu_trace_server_retire: asks a live launch to leave and returns when it has (a few tens of microseconds). */
int mgpu_trace_server_stats(MgpuScene *scene, uint64_t *launches, uint64_t *calls, int *alive, double *device_us);
int mgpu_trace_server_retire(MgpuScene *scene);
/* The same with rays, records and hit flags resident in device memory (d_out 16-byte aligned), enqueued on `stream`
 * (a hipStream_t, NULL = default stream) without synchronising; with stats != NULL the call waits for the kernel and
 * returns its counters and time. */
int mgpu_trace_device(MgpuScene *scene, const MgpuRay *d_rays, size_t n, MgpuIntersection *d_out, uint8_t *d_hit,
                      void *stream, MgpuStats *stats);
/* Threading contract of the *_device entry points: calls for one scene come from one host thread at a time (the
 * host-buffer entry points mgpu_trace / mgpu_render / mgpu_render_step / mgpu_render_panoramic take a lock and may be
 * called concurrently).  mgpu_render_strips_device keeps its launch scratch per stream (see below);
 * mgpu_trace_device and mgpu_render_panoramic_device share one set per scene (stack overflow columns of deep trees,
 * counters, events): successive calls must use ONE stream, or be separated by a synchronisation. */

/* -- Render (render.cc:593-708: per-pixel PathTrace, render.cc:381-456) ---------------------------------------------- */
/* Renders `passes` passes of the window [x0,x1) x [y0,y1) of a W x H frame and returns, per window pixel, the float32
 * sum over the passes of the per-pass radiance, added in pass order (== Render() followed by AccumImage,
 * main_sdl.cc:138-143; with passes == 1 exactly what Render() stores).
 *   origin/corner/du/dv : the camera frame of Camera::BuildCameraFrame (camera.cc:40-220)
 *   maxPathLength       : the reference's kMaxPathLength (render.cc:52); bounces = maxPathLength - 1
 *   plane               : {a,b,c,d} of the debug ground plane (prim-plane.h:15-24) or NULL when config.plane is false
 *   image_out           : HOST buffer, full-frame indexing 3*(y*W+x)+c; only window pixels are written
 *   count_out           : HOST buffer W*H or NULL; count[y*W+x] += passes for window pixels (render.cc:677-679)
 */
int mgpu_render(MgpuScene *scene, const double origin[3], const double corner[3], const double du[3],
                const double dv[3], int W, int H, int x0, int y0, int x1, int y1, int maxPathLength, int passes,
                const float plane[4], int rng_mode, const uint32_t *rng_states, uint64_t seed, uint32_t pass_base,
                float *image_out, int32_t *count_out, MgpuStats *stats);

/* Render() in the reference's OWN random stream (render.cc:116-168 with one OpenMP thread: every pixel continues the
 * xorshift128 state the previous pixel left, pass after pass): `passes` consecutive Render() calls on the whole frame,
 * summed in pass order as mgpu_render does.  stream_state = the generator's 4 words, IN (123456789, 362436069, 521288629,
 * 88675123 for a fresh process, render.cc:123-127) and OUT (where the next call continues).  The chain is resolved on the
 * device (mgpu_stream.hip: a pixel consumes 2 or 2 + 3 (maxPathLength - 1) draws depending on whether its primary ray hits,
 * so start states follow from the primary hit flags of all earlier pixels; one workgroup settles them by speculation), then
 * the frame is rendered from the resulting start-state table.  With the default seed the image is the reference's image.
 * states_out (nullable): the passes * W * H * 4 words of that table (MGPU_RNG_TABLE layout).
 * The table lives on the device for the duration of the call: 16 bytes per (pass, pixel), i.e. 33 MB per 1080p pass (the
 * reference renders one pass per Render() call).  One lock is held over both phases, and stream_state is advanced only when
 * the frame has been rendered: a failed call leaves the reference stream where it was. */
int mgpu_render_stream(MgpuScene *scene, const double origin[3], const double corner[3], const double du[3],
                       const double dv[3], int W, int H, int maxPathLength, int passes, const float plane[4],
                       uint32_t stream_state[4], float *image_out, int32_t *count_out, uint32_t *states_out, MgpuStats *stats);

/* What the last mgpu_render_stream call's resolution of the reference's random stream took (wall ms, all passes of the call, from
 * the first kernel to the verdict of its verification), whether it had to classify the camera's pixels first (a camera's classes
 * are cached with the scene), how many resolutions of this scene had to be repeated because a pixel classified "certain" was
 * not, and how many pixels of a pass are uncertain (silhouette pixels of mesh + plane against the sky). */
/* Render-ahead for progressive callers of mgpu_render (off by default; mallie::Render switches it on): a call that renders whole
 * rows with MGPU_RNG_HASH and stats == NULL enqueues, before it copies its own frame to the caller, the frame the NEXT call will
 * ask for if it repeats the arguments with pass_base moved on by `passes` -- what render.cc's drivers do pass after pass -- so
 * that kernel runs under this call's PCIe copy.  A next call that asks for exactly that finds it done; any other call drops it.
 * Images do not depend on it.  mgpu_render_ahead_stats: calls served from a frame rendered ahead / calls that were not. */
int mgpu_scene_set_render_ahead(MgpuScene *scene, int on);
int mgpu_render_ahead_stats(MgpuScene *scene, unsigned long long *hits, unsigned long long *misses);
int mgpu_stream_stats(MgpuScene *scene, double *resolve_ms, int *classified, unsigned long long *retries, uint32_t *uncertain_pixels);
/* Render() with its `step` argument (render.cc:657-696): step == 1 is mgpu_render with passes = 1 on the whole frame.
 * step > 1 traces one path per step x step block -- the path of the block's top-left pixel (its jitter, its RNG start
 * state: TABLE index / HASH pixel id are those of that pixel in the W x H frame) -- and fills the block with its radiance;
 * count_out is incremented by 3 per pixel, as the reference's fill loop does (once per colour channel).  W and H must be
 * multiples of step: otherwise the reference writes outside the image, and this returns MGPU_ERR_UNSUPPORTED. */
int mgpu_render_step(MgpuScene *scene, const double origin[3], const double corner[3], const double du[3],
                     const double dv[3], int W, int H, int step, int maxPathLength, const float plane[4], int rng_mode,
                     const uint32_t *rng_states, uint64_t seed, uint32_t pass_base, float *image_out, int32_t *count_out,
                     MgpuStats *stats);

/* ShowNormal / ShowUV (render.cc:458-516), the reference's two debug integrators (no caller there; they share Render()'s
 * per-pixel prologue): one jittered primary ray per pixel of the whole frame, Scene::Trace against the mesh only, and the
 * pixel is the hit's shading normal * 0.5 + 0.5 (MGPU_AOV_NORMAL) or (0.1 * texcoord[0], 0, 0) (MGPU_AOV_UV; a mesh without
 * facevarying_uvs gives 0 -- the reference reads an unset field there); black on a miss.  image_out (3*W*H float32, host)
 * is overwritten, count_out (nullable) incremented by 1.  RNG modes as mgpu_render, one start state per pixel. */
enum { MGPU_AOV_NORMAL = 0, MGPU_AOV_UV = 1 };
int mgpu_render_aov(MgpuScene *scene, const double origin[3], const double corner[3], const double du[3], const double dv[3],
                    int W, int H, int kind, int rng_mode, const uint32_t *rng_states, uint64_t seed, uint32_t pass_base,
                    float *image_out, int32_t *count_out, MgpuStats *stats);

/* Device-resident variant used by the multi-GPU strip renderer and the benchmark: nothing crosses PCIe.
 * The pixel set is a list of row strips: local row j (0 <= j < n_rows) is frame row
 *     y = y_first + (j / strip_h) * y_period + (j % strip_h)
 * and columns [x0,x1).  d_image (DEVICE pointer, 3*n_rows*(x1-x0) floats, row-major over local rows) receives the
 * pass-ordered float32 sum; d_count (DEVICE pointer or NULL, n_rows*(x1-x0) int32) is incremented by `passes`.
 * d_rng_states is a DEVICE pointer in the MGPU_RNG_TABLE layout (full-frame indexing) or NULL.
 * `stream` is a hipStream_t (NULL = the default stream); the call is asynchronous unless stats != NULL.
 * Launches on ONE stream run in order.  Launches on DIFFERENT streams of the same scene may overlap on the device --
 * the scene keeps one set of launch scratch (pass planes, work counters, tile order) per stream, four sets at most; a
 * fifth stream re-uses the least recently used set behind an event, i.e. it is ordered after that set's last launch.
 * Keeping two or three frames in flight this way fills the drain of one persistent launch (its last paths finishing
 * on a mostly idle GPU) with the next frame's work: 6.50 -> 6.32 ms per C2 frame, 1.28 -> 1.01 ms per eighth of it.
 * Calls for one scene must still come from one host thread at a time; a call with stats != NULL resets and reads the
 * scene's counters and should not overlap other launches. */
int mgpu_render_strips_device(MgpuScene *scene, const double frame[12], int W, int H, int x0, int x1, int y_first,
                              int strip_h, int y_period, int n_rows, int maxPathLength, int passes,
                              const float plane[4], int rng_mode, const uint32_t *d_rng_states, uint64_t seed,
                              uint32_t pass_base, float *d_image, int32_t *d_count, void *stream, MgpuStats *stats);
/* n_frames consecutive frames of the same window in one call: frame f renders passes pass_base + f * passes ... into
 * d_images[f] (and d_counts[f]; d_counts may be NULL) -- bit for bit what n_frames calls of mgpu_render_strips_device with
 * pass_base advancing by `passes` produce.  As many frames as the scratch budget holds (8 GiB of per-pass planes unless
 * MGPU_PLANES_MAX_MB says otherwise) share ONE persistent launch, so the end of a launch -- waves running out of work one
 * after the other -- is paid once per launch and not once per frame: rank 0's eighth of the 1080p Cornell frame takes 1.10
 * ms alone and 0.80 ms as one of four.  MGPU_RNG_TABLE: d_rng_states holds n_frames * passes tables, frame-major.  `stats`
 * (if not NULL) covers all frames; kernel_ms then includes the per-frame sums of the planes. */
int mgpu_render_frames_device(MgpuScene *scene, const double frame[12], int W, int H, int x0, int x1, int y_first,
                              int strip_h, int y_period, int n_rows, int maxPathLength, int passes,
                              const float plane[4], int rng_mode, const uint32_t *d_rng_states, uint64_t seed,
                              uint32_t pass_base, int n_frames, float *const *d_images, int32_t *const *d_counts,
                              void *stream, MgpuStats *stats);

/* -- Multi-GPU frames: the image is cut into interleaved strips of `strip_h` rows, rank r owns strips
 *    r, r + world, ...; the scene is replicated (one MgpuScene per GPU, created by the caller with mgpu_scene_create on
 *    that device); every GPU renders its strips (Render() semantics, MGPU_RNG_HASH seeding, so the frame does not depend on
 *    the GPU count) and ONE exchange step per frame -- grouped ncclSend / ncclRecv over RCCL / xGMI, each strip received
 *    at its final rows -- assembles the float RGB frame in rank 0's HBM.  RCCL is loaded on first use (dlopen), and not
 *    at all for one GPU.  Up to 16 frames may be in flight (own streams and buffers each): mgpu_frame_render only
 *    enqueues, mgpu_frame_wait blocks until a slot's frame is complete. -------------------------------------------------- */
typedef struct MgpuFrame MgpuFrame;
/* How the strips travel to rank 0 (MGPU_FRAME_EXCHANGE=block|strips when the frame is created; default block):
 *   MGPU_EXCHANGE_BLOCK   a rank's strip buffer is ONE message into a staging area on rank 0; one strided device copy per
 *                         rank deals it to the strips' final rows (world - 1 receives + as many 2-D copies per frame)
 *   MGPU_EXCHANGE_STRIPS  one send / receive pair per strip, received at its final rows (no staging; 118 pairs per 1080p
 *                         frame at eight ranks, 236 at 3840x2160) */
enum { MGPU_EXCHANGE_BLOCK = 0, MGPU_EXCHANGE_STRIPS = 1 };
/* What carries the bytes (MGPU_FRAME_TRANSPORT=rccl|copy when the frame is created; default rccl): with `copy`, and only for a
 * frame whose ranks are all driven by this process (mgpu_frame_create), every send / receive pair of the exchange step is a
 * device-to-device copy on rank 0's communicator stream instead -- same plan, staging, placement and events.  Nothing in it
 * needs one GPU per rank, so devices[] may then name a device several times: the N > 1 machinery for N = 2 .. 8 on a box with
 * one GPU (tests).  mgpu_frame_stats reports exchange_mode + 2 for it. */
/* One process driving n GPUs (ncclCommInitAll): scenes[r] must live on devices[r] (MGPU_ERR_INVALID otherwise) and becomes
 * rank r. */
int mgpu_frame_create(MgpuScene *const *scenes, const int *devices, int n, int W, int H, int strip_h, int frames_in_flight,
                      MgpuFrame **out);
/* One process per GPU (ncclCommInitRank): `id` = the 128 bytes mgpu_frame_unique_id produced on ONE rank, distributed by the
 * caller (MPI, torch.distributed, a file ...); may be NULL when world == 1.  Collective: every rank must call it. */
int mgpu_frame_unique_id(unsigned char id[128]);
int mgpu_frame_create_rank(MgpuScene *scene, int device, int rank, int world, const unsigned char id[128], int W, int H,
                           int strip_h, int frames_in_flight, MgpuFrame **out);
int mgpu_frame_destroy(MgpuFrame *frame);
/* Enqueues one frame of `passes` passes (camera frame as mgpu_camera_frame; plane NULL = off; rng_mode MGPU_RNG_HASH) on
 * the next slot and returns that slot's index.  Collective across the ranks of a multi-process frame.  A render call that
 * fails leaves the frame object unusable (work already enqueued on some GPUs and not on others): every later call on it
 * except mgpu_frame_destroy returns MGPU_ERR_INVALID. */
int mgpu_frame_render(MgpuFrame *frame, const double cam[12], int maxPathLength, int passes, const float plane[4],
                      int rng_mode, uint64_t seed, uint32_t pass_base, int *slot_out);
/* n_frames (<= frames_in_flight) consecutive frames -- frame i renders passes pass_base + i * passes ... -- rendered by ONE
 * launch per GPU (mgpu_render_frames_device) and exchanged frame by frame; slots_out[i] (may be NULL) receives frame i's
 * slot.  Same frames as n_frames calls of mgpu_frame_render.  Collective like it. */
int mgpu_frame_render_batch(MgpuFrame *frame, const double cam[12], int maxPathLength, int passes, const float plane[4],
                            int rng_mode, uint64_t seed, uint32_t pass_base, int n_frames, int *slots_out);
/* Waits for the frame of `slot`.  On the process that holds rank 0: *device_image (nullable) receives the device pointer
 * of the H x W x 3 float frame (valid until the slot is used again), host_image (nullable) a copy of it. */
int mgpu_frame_wait(MgpuFrame *frame, int slot, float *host_image, float **device_image);
/* SURVEY 8(d)'s frame ends with one read-back (render.cc:673-679 writes the caller's host image).  With the read-back switched
 * on, every slot of rank 0's process owns a pinned host buffer and mgpu_frame_wait_host(slot) -- once the slot's frame is
 * complete -- copies the frame into it on a copy stream of its own, waits for the copy and hands the buffer out (H x W x 3 floats,
 * valid until the slot's next render call).  The intended use is a pipeline: enqueue frame k + 1 (mgpu_frame_render), THEN take
 * frame k -- its 24.9 MB (1080p) then cross PCIe under frame k + 1's kernel and cost nothing (needs frames_in_flight >= 2; a
 * slot is not rendered into again before its copy has left it).  On one GPU the frames of a frame object with the read-back on
 * go down ONE stream, in order (two whole-GPU launches enqueued on two streams share the CUs and finish together, which leaves
 * nothing to hide the first copy under).  With several GPUs the copy is enqueued by the render call itself, behind the frame's
 * exchange (rank 0 renders an N-th of a frame while a whole frame crosses PCIe: the copies run as the frames arrive), and
 * mgpu_frame_wait_host only waits for it.  A frame that is not taken before its slot is rendered into again is lost: the slot then
 * holds the newer frame, and that is what a later mgpu_frame_wait_host hands out.  In a process that does not hold rank 0, mgpu_frame_wait_host waits until the slot's
 * strips have left and returns MGPU_OK with *host_image = NULL. */
int mgpu_frame_set_readback(MgpuFrame *frame, int on);
int mgpu_frame_wait_host(MgpuFrame *frame, int slot, const float **host_image);
/* What the frame object knows about itself and its exchange step: `rccl_ranks` is read back from the communicator
 * (ncclCommCount; 0 when no communicator exists, i.e. one GPU without MGPU_FRAME_FORCE_EXCHANGE); `exchange_ms` is the device
 * time of the exchange steps (grouped sends / receives + the placement copies) of `exchange_frames` frames, from HIP events
 * on rank 0's communicator stream -- only on the process that holds rank 0, and only for frames whose slot was waited for or
 * reused after their exchange had finished.  Waits for exchanges still in flight.  reset != 0 zeroes the sums. */
typedef struct {
  int world, members, rccl_ranks, exchange_mode;
  uint64_t frames;                 /* frames enqueued so far                                  */
  uint64_t exchange_frames;        /* frames whose exchange step was timed                    */
  uint64_t exchange_ops_per_frame; /* receives rank 0 posts per frame                         */
  double exchange_ms;              /* summed over exchange_frames                             */
  uint64_t enqueue_calls;          /* render calls (mgpu_frame_render / _render_batch) since the last reset               */
  double enqueue_ms;               /* host time those calls spent enqueueing (launches, events, the exchange) for ALL members */
} MgpuFrameStats;
int mgpu_frame_stats(MgpuFrame *frame, MgpuFrameStats *out, int reset);
/* Makes `stream` (a hipStream_t on rank 0's device) wait for the slot's frame without blocking the host. */
int mgpu_frame_done_event_wait(MgpuFrame *frame, int slot, void *stream);
/* Rows of a W x H frame that rank `rank` of `world` owns with strips of strip_h rows (-1 on bad arguments). */
int mgpu_frame_rows(int H, int strip_h, int world, int rank);
/* The exchange plan of rank `owner`: one piece per strip, in strip order -- offset in the owner's local strip buffer,
 * offset in the whole frame, number of floats.  The sender walks it with ncclSend, rank 0 with ncclRecv.  Returns the
 * number of pieces (the arrays receive at most max_pieces of them; any may be NULL), -1 on bad arguments. */
int mgpu_frame_plan(int W, int H, int strip_h, int world, int owner, size_t *local_off, size_t *frame_off, size_t *count,
                    int max_pieces);
/* The block exchange of rank `owner` as numbers (host-only; the device code uses the same function): out[0] = offset (floats)
 * of its strip buffer in rank 0's staging area, out[1] = floats of the one message it sends, out[2..6] = dst offset, dst pitch,
 * src pitch, width (all bytes) and height (rows) of the strided 2-D copy that deals its full strips to their rows of the frame,
 * out[7..9] = dst offset, src offset and bytes of the plain copy of a partial last strip (0 bytes: none).  Rank 0's own strips
 * are placed with the same numbers straight from its strip buffer (no message) unless force_exchange != 0.  -1 on bad
 * arguments. */
int mgpu_frame_block_plan(int W, int H, int strip_h, int world, int owner, int force_exchange, size_t out[10]);
const char *mgpu_frame_last_error(void);

/* -- RenderPanoramic (render.cc:710-763; PathTraceEnv render.cc:518-590; Camera::GenerateEnvRay / GenerateStereoEnvRay
 *    camera.cc:242-329) -- what the reference's console driver renders (main_console.cc:111) --------------------------- */
/* One RenderPanoramic() call on the window [x0,x1) x [y0,y1) of a W x H equirectangular frame: every window pixel of
 * `image_out` (3*W*H float32, full-frame indexing) is overwritten with the sum of `samples` (reference: 10) PathTraceEnv
 * radiances, added as `float += double` sample after sample; count_out[px] += samples (may be NULL).
 * origin = the camera frame's origin (mgpu_camera_frame); stereo != 0 selects GenerateStereoEnvRay (left eye in the top
 * half of the frame); maxPathLength: the reference's kMaxPathLength, 16.
 * A pixel's samples draw from ONE xorshift128 stream, so the RNG modes are per PIXEL: MGPU_RNG_TABLE reads the state at
 * the pixel's first sample from rng_states[(y*W + x)*4 ..] (capture them from a reference-stream run to reproduce its
 * image), MGPU_RNG_HASH uses mgpu_hash_state(seed, pass_base, y*W + x).  MGPU_RNG_STREAM is refused as in mgpu_render. */
int mgpu_render_panoramic(MgpuScene *scene, const double origin[3], int W, int H, int x0, int y0, int x1, int y1,
                          int maxPathLength, int samples, int stereo, int rng_mode, const uint32_t *rng_states,
                          uint64_t seed, uint32_t pass_base, float *image_out, int32_t *count_out, MgpuStats *stats);
/* The same with device buffers, asynchronous on `stream`: d_image (3 * (x1-x0) * (y1-y0) floats) and d_count are
 * WINDOW-local, d_rng_states (TABLE mode) covers the full frame.  With stats != NULL the call waits for the kernel. */
int mgpu_render_panoramic_device(MgpuScene *scene, const double origin[3], int W, int H, int x0, int y0, int x1, int y1,
                                 int maxPathLength, int samples, int stereo, int rng_mode, const uint32_t *d_rng_states,
                                 uint64_t seed, uint32_t pass_base, float *d_image, int32_t *d_count, void *stream,
                                 MgpuStats *stats);

/* Device work counters accumulate over every mgpu_render* call made with stats == NULL (calls with stats != NULL zero
 * them first and return that call's own counts).  mgpu_stats_read synchronises the device and returns the running
 * totals (kernel_ms / total_ms are left 0); reset != 0 zeroes them afterwards.  With the render-ahead on
 * (mgpu_scene_set_render_ahead) the totals include the frames rendered ahead, served or dropped. */
int mgpu_stats_read(MgpuScene *scene, MgpuStats *out, int reset);

/* Per-launch kernel timing for asynchronous use: after mgpu_timing_enable(scene, 1) every mgpu_render_strips_device call
 * made with stats == NULL brackets its kernel with HIP events on the launch stream (no synchronisation).
 * mgpu_timing_read synchronises, returns the summed kernel time and the number of launches since the last read, and
 * rearms the event ring (at most 4096 launches between reads). */
int mgpu_timing_enable(MgpuScene *scene, int on);
int mgpu_timing_read(MgpuScene *scene, double *total_ms, int *launches);

/* Diagnostic: traces ONE eye path (pixel px,py, given 128-bit start state) on the device and returns one record of 16
 * doubles per PathTrace loop iteration actually executed (render.cc:402-453): org[3], dir[3], t, hit (0/1), BVH slot
 * of the mesh hit or -1 (plane / miss), shading normal[3], Intersection::materialID, pathLength, throughput.x and
 * radiance.x BEFORE the iteration's update. records must hold 16*maxPathLength doubles. */
int mgpu_probe_path(MgpuScene *scene, const double frame[12], int W, int H, int px, int py, int maxPathLength,
                    const float plane[4], const uint32_t start_state[4], double *records, int *n_records);

/* Display transforms of the reference's drivers, fused with the per-pixel 1/count (device pointers, asynchronous):
 *   MGPU_TONEMAP_LINEAR_RGB8   HDRToLDR + fclamp, main_console.cc:25-43: 3 bytes/pixel, clamp(int((in/count) * 255.5))
 *   MGPU_TONEMAP_GAMMA22_BGRA8 Display + fclamp, main_sdl.cc:157-165,420-477: 4 bytes/pixel B,G,R,255, gamma 2.2 */
enum { MGPU_TONEMAP_LINEAR_RGB8 = 0, MGPU_TONEMAP_GAMMA22_BGRA8 = 1 };
int mgpu_tonemap_device(int device, const float *d_image, const int32_t *d_count, size_t npix, int mode,
                        unsigned char *d_out, void *stream);

/* The per-(pixel,pass) start state of MGPU_RNG_HASH (host helper; the device uses the same function). */
void mgpu_hash_state(uint64_t seed, uint32_t pass, uint32_t pixel, uint32_t state[4]);

/* -- host-side pieces of the path (plain C++ on the CPU, as in the reference; no device work) -------------------------- */
/* Camera::BuildCameraFrame (camera.cc:40-220): frame = origin[3], corner[3], du[3], dv[3]. */
int mgpu_camera_frame(const double eye[3], const double lookat[3], const double up[3], const double quat[4], double fov,
                      int width, int height, double frame[12]);
/* BVHAccel::Build (bvh_accel.cc:445-482) with explicit BVHBuildOptions; *nodes_out / *indices_out are malloc'ed, release
 * with mgpu_free. stats = {maxTreeDepth, numLeafNodes, numBranchNodes}. */
int mgpu_bvh_build(const double *verts, size_t nv, const uint32_t *faces, size_t nf, double costTaabb,
                   int minLeafPrimitives, int maxTreeDepth, int binSize, MgpuNode **nodes_out, size_t *nn_out,
                   uint32_t **indices_out, int stats[3]);
/* The same build ON THE DEVICE: a breadth-first parallel evaluation of the reference algorithm that
 * returns byte-identical nodes and indices (including libstdc++'s std::partition element order); host arrays in and
 * out, released with mgpu_free. binSize <= 256. device_ms (nullable) receives the device time of the build kernels. */
int mgpu_bvh_build_device(const double *verts, size_t nv, const uint32_t *faces, size_t nf, double costTaabb,
                          int minLeafPrimitives, int maxTreeDepth, int binSize, int device, MgpuNode **nodes_out,
                          size_t *nn_out, uint32_t **indices_out, int stats[3], double *device_ms);
void mgpu_free(void *p);
/* The debug ground plane Render() derives from the scene box on its first call (render.cc:620-627). */
void mgpu_plane_from_bbox(const double bmin[3], const double bmax[3], float plane[4]);

#ifdef __cplusplus
}
#endif
#endif /* MGPU_H_ */
