/*
 * oracle/mallie_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, CPU, fp64 restatement of the render hot path of lighttransport/mallie
 * (render.cc::Render/PathTrace -> Scene::Trace -> BVHAccel::Traverse, plus the host-side pieces that
 * determine its inputs: BVHAccel::Build, Camera::BuildCameraFrame, Plane::intersect).
 *
 * It exists to CHECK the HIP product path (mallie_amd/csrc) and to serve as the timed CPU baseline in
 * bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement bit-for-bit against vectors
 * produced by the unmodified reference sources compiled from /root/reference (oracle/Makefile target `ref`,
 * oracle/ref_driver.cc, oracle/make_goldens.py): camera frames, the BVH (nodes + indices), (Ray ->
 * Intersection) batches and Render() images for 1 and 2 consecutive passes, plane on and off.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction, no fast-math) -- see oracle/Makefile.
 */
#ifndef MALLIE_ORACLE_H_
#define MALLIE_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 64-byte node, the reference's BVHNode layout (bvh_accel.h:10-30). */
typedef struct {
  double bmin[3];
  double bmax[3];
  int32_t flag; /* 1 = leaf, 0 = branch */
  int32_t axis; /* split axis (branch); the reference leaves it uninitialised for leaves, we store 0 */
  uint32_t data[2]; /* leaf: count, first index; branch: child0, child1 */
} mo_node;

/* Result record of one Scene::Trace call (the fields of intersection.h:6-24 the mesh path writes). */
typedef struct {
  uint32_t hit;
  uint32_t faceID;
  uint32_t materialID;
  uint32_t f0, f1, f2;
  double t, u, v;
  double position[3];
  double geometricNormal[3];
  double normal[3];
  double texcoord[2];
} mo_hit; /* 24 + 14*8 = 136 bytes */

/* Work counters (all in units of one event). */
typedef struct {
  uint64_t trace_calls; /* reference-equivalent Scene::Trace() calls                           */
  uint64_t real_rays;   /* Trace() calls made before (and including) the path's first miss      */
  uint64_t nodes;       /* BVH nodes popped + box-tested by real rays                           */
  uint64_t tris;        /* triangle tests by real rays                                          */
  uint64_t garbage_nodes; /* nodes popped by post-miss continuation rays            */
  uint64_t garbage_hits;  /* post-miss continuation rays that hit anything (expected 0)         */
  uint64_t paths;       /* eye paths started (= pixels * passes)                                */
  uint64_t max_stack;   /* deepest traversal stack index seen + 1                               */
} mo_stats;

enum { MO_RNG_STREAM = 0, MO_RNG_TABLE = 1, MO_RNG_HASH = 2 };

typedef struct mo_scene mo_scene;

/* ---- ShowNormal / ShowUV (render.cc:458-516) ------------------------------------------------------------ */
int mo_render_aov(const mo_scene *s, const double frame[12], int W, int H, int mode, int rng_mode, uint32_t stream_state[4],
                  const uint32_t *rng_states, uint64_t seed, uint32_t pass_base, float *image, uint32_t *states_out,
                  mo_stats *stats);

/* ---- Render(step) ------------------------------------------------------------------------------------ */
/* One Render() call with its `step` argument (render.cc:657-696); see mallie_oracle.c. Returns 0 on success. */
int mo_render_step(const mo_scene *s, const double frame[12], int W, int H, int step, int maxPathLength,
                   const float *plane, int rng_mode, uint32_t stream_state[4], const uint32_t *rng_states, uint64_t seed,
                   uint32_t pass_base, float *image, int32_t *count, uint32_t *states_out, mo_stats *stats);

/* ---- BVH build (bvh_accel.cc:36-482) ------------------------------------------------------------ */
/* Returns 0 on success. *nodes_out / *indices_out are malloc'ed; release with mo_free. stats = {maxTreeDepth,
 * numLeafNodes, numBranchNodes}. */
int mo_bvh_build(const double *verts, size_t nv, const uint32_t *faces, size_t nf, double costTaabb,
                 int minLeafPrimitives, int maxTreeDepth, int binSize, mo_node **nodes_out, size_t *nn_out,
                 uint32_t **indices_out, int stats[3]);
void mo_free(void *p);

/* ---- scene ---------------------------------------------------------------------------------------- */
/* All arrays are copied. fv_normals (9*nf) and fv_uvs (6*nf) may be NULL; matIDs may be NULL (then every hit gets
 * materialID 0xFFFFFFFF as bvh_accel.cc:687-691). mat_diffuse holds 3*nm doubles (nm may be 0: every lookup then
 * yields the default Material().diffuse = 0.5, scene.h:58-65). */
mo_scene *mo_scene_create(const double *verts, size_t nv, const uint32_t *faces, size_t nf, const uint32_t *matIDs,
                          const double *fv_normals, const double *fv_uvs, const mo_node *nodes, size_t nn,
                          const uint32_t *indices, const double *mat_diffuse, size_t nm);
void mo_scene_destroy(mo_scene *s);
/* Scene::BoundingBox (scene.cc:317-333) = root node box. */
void mo_scene_bbox(const mo_scene *s, double bmin[3], double bmax[3]);
/* Plane coefficients exactly as Render() derives them on its first call (render.cc:620-627). */
void mo_plane_from_bbox(const double bmin[3], const double bmax[3], float plane[4]);

/* ---- Scene::Trace, batched (scene.cc:253, bvh_accel.cc:773-844) ----------------------------------- */
/* rays: 6 doubles each (org, dir). stats may be NULL. */
int mo_trace(const mo_scene *s, const double *rays, size_t n, mo_hit *out, mo_stats *stats);

/* ---- camera (camera.cc:40-240) -------------------------------------------------------------------- */
/* frame = origin[3], corner[3], du[3], dv[3]. */
void mo_camera_frame(const double eye[3], const double lookat[3], const double up[3], const double quat[4], double fov,
                     int width, int height, double frame[12]);
void mo_generate_ray(const double frame[12], double u, double v, double ray[6]);

/* ---- RNG ------------------------------------------------------------------------------------------ */
/* xorshift128 step (render.cc:137-168); returns w * 2^-32. */
double mo_xorshift128(uint32_t st[4]);
/* Per-(pixel, pass) start state for MO_RNG_HASH: two splitmix64 outputs of a counter built from
 * (seed, pass, pixel); never all-zero. Shared definition with the HIP path (mallie_amd/csrc/mgpu_device.h). */
void mo_hash_state(uint64_t seed, uint32_t pass, uint32_t pixel, uint32_t st[4]);

/* ---- Render (render.cc:593-708 + PathTrace render.cc:381-456) ------------------------------------- */
/*
 * Renders `passes` consecutive passes of the window [x0,x1) x [y0,y1) of a W x H image.
 *   image : 3*W*H float32, full-frame indexing; for every pixel of the window it receives the SUM over the passes
 *           of the per-pass float radiance, added in pass order in float32 (== Render() + AccumImage,
 *           main_sdl.cc:138-143). With passes == 1 this is exactly what Render() leaves in `image`.
 *           Pixels outside the window are not touched; pixels inside are overwritten (not accumulated into).
 *   count : W*H int32, count[px] += passes for window pixels (render.cc:677-679). May be NULL.
 *   plane : 4 floats (a,b,c,d) or NULL for "plane": false.
 *   maxPathLength : the reference's kMaxPathLength (16).  bounces = maxPathLength - 1.
 *   rng_mode :
 *     MO_RNG_STREAM  reference stream: ONE xorshift128 state (stream_state, in/out; the reference's thread-0 seed is
 *                    {123456789,362436069,521288629,88675123}) consumed in scanline order exactly as the reference
 *                    does with OMP_NUM_THREADS=1.  Serial.  Only meaningful for the full window.
 *     MO_RNG_TABLE   per-(pass,pixel) start states: rng_states[((pass*H + y)*W + x)*4 .. +4].
 *     MO_RNG_HASH    per-(pass,pixel) start states from mo_hash_state(seed, pass_base + pass, y*W + x).
 *   states_out : if non-NULL (any mode), receives the start state of every (pass,pixel) in the MO_RNG_TABLE layout --
 *                this is how a reference-stream run is turned into a table the GPU can replay.
 *   nthreads : OpenMP threads for TABLE/HASH modes (rows are independent there); STREAM ignores it.
 * Returns 0 on success.
 */
int mo_render(const mo_scene *s, const double frame[12], int W, int H, int x0, int y0, int x1, int y1,
              int maxPathLength, int passes, const float *plane, int rng_mode, uint32_t stream_state[4],
              const uint32_t *rng_states, uint64_t seed, uint32_t pass_base, float *image, int32_t *count,
              uint32_t *states_out, mo_stats *stats, int nthreads);

/* ---- RenderPanoramic (render.cc:710-763 + PathTraceEnv render.cc:518-590 + camera.cc:242-329) ---------------------- */
/* Camera::GenerateEnvRay (stereo == 0) / GenerateStereoEnvRay: ray[0..2] = origin, ray[3..5] = direction. */
void mo_generate_env_ray(const double origin[3], int W, int H, int stereo, double u, double v, double ray[6]);
/*
 * One RenderPanoramic() call on the window [x0,x1) x [y0,y1): every pixel receives the sum of `samples` (reference: 10)
 * PathTraceEnv radiances, added one after the other as `float += double`; count[px] += samples.
 * A pixel's samples draw from ONE xorshift128 stream (the reference's loop nest is y, x, sample), so the RNG modes are
 * per PIXEL: MO_RNG_STREAM = one state for the whole scanline-ordered frame (in/out), MO_RNG_TABLE = rng_states[px*4..]
 * is the state at the pixel's first sample, MO_RNG_HASH = mo_hash_state(seed, pass_base, px).  states_out (W*H*4) receives
 * every pixel's start state.  maxPathLength: the reference's 16.
 */
int mo_render_panoramic(const mo_scene *s, const double origin[3], int W, int H, int x0, int y0, int x1, int y1,
                        int maxPathLength, int samples, int stereo, int rng_mode, uint32_t stream_state[4],
                        const uint32_t *rng_states, uint64_t seed, uint32_t pass_base, float *image, int32_t *count,
                        uint32_t *states_out, mo_stats *stats, int nthreads);

/* Display transforms: mode 0 = HDRToLDR/fclamp of main_console.cc:25-43 (RGB8, linear), mode 1 = Display/fclamp of
 * main_sdl.cc:157-165,420-477 (BGRA8, gamma 2.2); both divide by the per-pixel count first. */
void mo_tonemap(const float *image, const int32_t *count, size_t npix, int mode, unsigned char *out);

/* One eye path with per-iteration records: see mallie_oracle.c. records: 16*maxPathLength doubles. */
int mo_probe_path(const mo_scene *s, const double frame[12], int px, int py, int maxPathLength, const float *plane,
                  const uint32_t start_state[4], double *records, int *n_records, double radiance[3]);

#ifdef __cplusplus
}
#endif
#endif /* MALLIE_ORACLE_H_ */
