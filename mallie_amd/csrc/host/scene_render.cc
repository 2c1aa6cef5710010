// host/scene_render.cc -- mallie::Scene and mallie::Render of the facade (host glue around the C ABI).
//
//   Scene::Init / InitFromArrays   scene.cc:66-251   (mesh load, scale / fit-to-[-1,1]^3, BVH build)
//   Scene::Trace / BoundingBox     scene.cc:253-333
//   Render                         render.cc:593-708 (camera frame, first-call plane setup, one pass, count++)
#include <chrono>
#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "../../../include/mallie/mallie_api.hpp"
#include "../../../include/mgpu.h"
#include "mesh_io.hpp"

namespace mallie {

// ---- Scene ------------------------------------------------------------------------------------------------------------
Scene::Scene() { memset(&mesh_, 0, sizeof(mesh_)); }
Scene::~Scene() { FreeMesh(); }

void Scene::FreeMesh() {
  accel_.ReleaseDevice();
  delete[] mesh_.vertices;
  delete[] mesh_.faces;
  delete[] mesh_.materialIDs;
  delete[] mesh_.facevarying_normals;
  delete[] mesh_.facevarying_uvs;
  memset(&mesh_, 0, sizeof(mesh_));
}

bool Scene::Init(const std::string &objFilename, const std::string &esonFilename,
                 const std::string &magicaVoxelFilename, const std::string &materialFilename, double sceneScale,
                 bool sceneFit) {
  (void)materialFilename; // accepted and ignored, as in the reference
  FreeMesh();
  materials_.clear();
  bool ok = false;
  if (!objFilename.empty()) {
    ok = mesh_io::LoadObj(mesh_, objFilename.c_str());
    printf(ok ? "Mallie:info\tmsg:Success to load .obj file [ %s ]\n" : "Mallie:err\tmsg:Failed to load .obj file [ %s ]\n",
           objFilename.c_str());
  } else if (!esonFilename.empty()) {
    ok = mesh_io::LoadESON(mesh_, esonFilename.c_str());
    printf(ok ? "Mallie:info\tmsg:Success to load .eson file [ %s ]\n" : "Mallie:err\tmsg:Failed to load .eson file [ %s ]\n",
           esonFilename.c_str());
  } else if (!magicaVoxelFilename.empty()) {
    ok = mesh_io::LoadMagicaVoxel(mesh_, materials_, magicaVoxelFilename.c_str());
    printf(ok ? "Mallie:info\tmsg:Success to load .vox file [ %s ]\n" : "Mallie:err\tmsg:Failed to load .vox file [ %s ]\n",
           magicaVoxelFilename.c_str());
  }
  if (!ok) {
    printf("Mallie:err\tmsg:Failed to load mesh\n");
    return false;
  }
  return Finish(sceneScale, sceneFit);
}

bool Scene::InitFromArrays(const real *vertices, size_t numVertices, const unsigned int *faces, size_t numFaces,
                           const unsigned int *materialIDs, const real *facevarying_normals,
                           const real *facevarying_uvs, const std::vector<Material> &materials, double sceneScale,
                           bool sceneFit) {
  if (!vertices || !faces || !numVertices || !numFaces) return false;
  FreeMesh();
  mesh_.numVertices = numVertices;
  mesh_.numFaces = numFaces;
  mesh_.vertices = new real[3 * numVertices];
  memcpy(mesh_.vertices, vertices, sizeof(real) * 3 * numVertices);
  mesh_.faces = new unsigned int[3 * numFaces];
  memcpy(mesh_.faces, faces, sizeof(unsigned int) * 3 * numFaces);
  mesh_.materialIDs = new unsigned int[numFaces];
  if (materialIDs) memcpy(mesh_.materialIDs, materialIDs, sizeof(unsigned int) * numFaces);
  else memset(mesh_.materialIDs, 0, sizeof(unsigned int) * numFaces); // 0 = default material (mesh_loader.cc:295-297)
  if (facevarying_normals) {
    mesh_.facevarying_normals = new real[9 * numFaces];
    memcpy(mesh_.facevarying_normals, facevarying_normals, sizeof(real) * 9 * numFaces);
  }
  if (facevarying_uvs) {
    mesh_.facevarying_uvs = new real[6 * numFaces];
    memcpy(mesh_.facevarying_uvs, facevarying_uvs, sizeof(real) * 6 * numFaces);
  }
  materials_ = materials;
  return Finish(sceneScale, sceneFit);
}

bool Scene::Finish(double sceneScale, bool sceneFit) {
  real *v = mesh_.vertices;
  const size_t nv = mesh_.numVertices;
  if (sceneFit) { // scene.cc:112-160: fit to [-1,1]^3, each step applied as its own rounding
    real lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (size_t i = 0; i < nv; i++)
      for (int k = 0; k < 3; k++) {
        lo[k] = std::min(lo[k], v[3 * i + k]);
        hi[k] = std::max(hi[k], v[3 * i + k]);
      }
    real inv[3];
    for (int k = 0; k < 3; k++) {
      const real ext = hi[k] - lo[k];
      inv[k] = (ext > 0.000001) ? (1.0 / ext) : ext;
    }
    printf("bmin = %f, %f, %f\n", lo[0], lo[1], lo[2]);
    printf("bmax = %f, %f, %f\n", hi[0], hi[1], hi[2]);
    printf("binv = %f, %f, %f\n", inv[0], inv[1], inv[2]);
    for (size_t i = 0; i < nv; i++)
      for (int k = 0; k < 3; k++) {
        real &c = v[3 * i + k];
        c -= lo[k];
        c *= inv[k];
        c -= 0.5;
        c *= 2.0;
      }
  } else {
    for (size_t i = 0; i < 3 * nv; i++) v[i] *= sceneScale; // scene.cc:164-168
  }
  BVHBuildOptions options;
  printf("  BVH build option:\n    # of leaf primitives: %d\n    SAH binsize         : %d\n", options.minLeafPrimitives,
         options.binSize);
  if (!accel_.Build(&mesh_, options)) return false;
  const BVHBuildStatistics st = accel_.GetStatistics();
  printf("  BVH statistics:\n    # of leaf   nodes: %d\n    # of branch nodes: %d\n  Max tree depth   : %d\n",
         st.numLeafNodes, st.numBranchNodes, st.maxTreeDepth);
  real3 bmin, bmax;
  BoundingBox(bmin, bmax);
  printf("  BVH bounding box:\n    bmin = (%f, %f, %f)\n    bmax = (%f, %f, %f)\n", bmin[0], bmin[1], bmin[2], bmax[0],
         bmax[1], bmax[2]);
  return true;
}

bool Scene::Trace(Intersection &isect, Ray &ray) { return accel_.Traverse(isect, &mesh_, ray); }

void Scene::BoundingBox(real3 &bmin, real3 &bmax) {
  const std::vector<BVHNode> &nodes = accel_.GetNodes();
  if (nodes.empty()) {
    bmin = real3(0, 0, 0);
    bmax = real3(0, 0, 0);
    return;
  }
  for (int k = 0; k < 3; k++) {
    bmin[k] = nodes[0].bmin[k];
    bmax[k] = nodes[0].bmax[k];
  }
}

MgpuScene *Scene::CreateDeviceScene(int device) {
  const std::vector<BVHNode> &nodes = accel_.GetNodes();
  const std::vector<unsigned int> &indices = accel_.GetIndices();
  if (nodes.empty() || !mesh_.vertices) return NULL;
  std::vector<double> diffuse;
  for (size_t i = 0; i < materials_.size(); i++)
    for (int k = 0; k < 3; k++) diffuse.push_back(materials_[i].diffuse[k]);
  MgpuScene *s = NULL;
  const int rc = mgpu_scene_create(mesh_.vertices, mesh_.numVertices, mesh_.faces, mesh_.numFaces, mesh_.materialIDs,
                                   mesh_.facevarying_normals, mesh_.facevarying_uvs,
                                   reinterpret_cast<const MgpuNode *>(&nodes[0]), nodes.size(), &indices[0],
                                   diffuse.empty() ? NULL : &diffuse[0], diffuse.size() / 3, device, &s);
  if (rc != MGPU_OK) {
    printf("Mallie:err\tmsg:GPU scene upload to device %d failed: %s\n", device, mgpu_last_error());
    return NULL;
  }
  return s;
}

real3 Scene::GetBackgroundRadiance(real3 &dir) {
  (void)dir;
  return real3(0.75, 0.75, 0.75); // scene.cc:335-338
}

// ---- RenderConfig (render.h:33-48) ----------------------------------------------------------------------------------------
RenderConfig::RenderConfig()
    : fov(45.0), width(512), height(512), scene_scale(1.0), scene_fit(false), plane(false), num_passes(10),
      num_photons(10000) {
  eye[0] = 0.0; eye[1] = 0.0; eye[2] = -5.0;
  lookat[0] = lookat[1] = lookat[2] = 0.0;
  up[0] = 0.0; up[1] = 1.0; up[2] = 0.0;
  quat[0] = quat[1] = quat[2] = quat[3] = 0.0;
}

// ---- Render ---------------------------------------------------------------------------------------------------------------
namespace {

// process-wide render state, mirroring the reference's file-scope globals (render.cc:113-116,615)
bool gInitialPass = true;
bool gPlane = false;
Plane gPlaneObject;
int gMaxPathLength = 16;
unsigned long long gSeed = 1;
unsigned int gPassCounter = 0;
const unsigned int *gRngTable = NULL;
bool gReferenceStream = false;                                             // SetRenderReferenceStream
unsigned int gStreamState[4] = {123456789u, 362436069u, 521288629u, 88675123u}; // init_randomreal(), render.cc:123-127

// MALLIE_GPUS=n: the frame object that spreads Render() over n GPUs, kept while scene and frame size stay the same
struct MultiGpu {
  const Scene *scene;
  int n, W, H;
  std::vector<MgpuScene *> scenes; // [0] belongs to the Scene, the others are ours
  MgpuFrame *frame;
  MultiGpu() : scene(NULL), n(0), W(0), H(0), frame(NULL) {}
  void release() {
    if (frame) mgpu_frame_destroy(frame);
    frame = NULL;
    for (size_t i = 1; i < scenes.size(); i++) mgpu_scene_destroy(scenes[i]);
    scenes.clear();
    scene = NULL;
  }
};
MultiGpu gMulti;

int wanted_gpus() {
  const char *e = getenv("MALLIE_GPUS");
  if (!e) return 1;
  int n = atoi(e);
  const int have = mgpu_device_count();
  const char *t = getenv("MGPU_FRAME_TRANSPORT"); // copy transport: ranks may share a device (include/mgpu.h; tests)
  if (n > have && !(t && !strcmp(t, "copy") && n <= 16)) n = have;
  return n < 1 ? 1 : n;
}

MgpuFrame *multi_frame(Scene &scene, int n, int W, int H) {
  if (gMulti.frame && gMulti.scene == &scene && gMulti.n == n && gMulti.W == W && gMulti.H == H &&
      gMulti.scenes[0] == scene.DeviceScene())
    return gMulti.frame;
  gMulti.release();
  MgpuScene *first = scene.DeviceScene();
  if (!first) return NULL;
  gMulti.scenes.push_back(first);
  // rank 0 is wherever the Scene's own device copy lives (MALLIE_DEVICE, or LOCAL_RANK under a launcher: bvh_build.cc); the
  // replicas go to the next devices round the ring, so every member's streams and buffers sit on its scene's device
  const int have = mgpu_device_count();
  const int dev0 = mgpu_scene_device(first);
  std::vector<int> devices(1, dev0);
  for (int k = 1; k < n; k++) {
    const int d = (dev0 + k) % (have > 0 ? have : 1);
    MgpuScene *r = scene.CreateDeviceScene(d);
    if (!r) {
      gMulti.release();
      return NULL;
    }
    gMulti.scenes.push_back(r);
    devices.push_back(d);
  }
  if (mgpu_frame_create(&gMulti.scenes[0], &devices[0], n, W, H, 8, 1, &gMulti.frame) != MGPU_OK) {
    printf("Mallie:err\tmsg:multi-GPU frame: %s\n", mgpu_frame_last_error());
    gMulti.release();
    return NULL;
  }
  gMulti.scene = &scene;
  gMulti.n = n;
  gMulti.W = W;
  gMulti.H = H;
  return gMulti.frame;
}

void plane_from_bbox(const double bmin[3], const double bmax[3], float pl[4]) {
  // render.cc:622-626: float zmin, float zsize, d = -(zmin - zsize * 0.0001f)
  const float zmin = (float)bmin[1];
  const float zsize = (float)(bmax[1] - bmin[1]);
  pl[0] = 0;
  pl[1] = 1;
  pl[2] = 0;
  pl[3] = -(zmin - zsize * 0.0001f);
}

} // namespace

void SetMaxPathLength(int n) { gMaxPathLength = n < 1 ? 1 : n; }
void SetRenderSeed(unsigned long long seed) {
  gSeed = seed;
  gPassCounter = 0;
}
void SetRenderRngTable(const unsigned int *states) { gRngTable = states; }
static int gFastMode = -1; // SetRenderFastMode; -1: not set, MALLIE_FAST decides
void SetRenderFastMode(bool on) { gFastMode = on ? 1 : 0; }
void SetRenderReferenceStream(bool on) {
  gReferenceStream = on;
  gStreamState[0] = 123456789u; gStreamState[1] = 362436069u; gStreamState[2] = 521288629u; gStreamState[3] = 88675123u;
}

static bool render_impl(Scene &scene, const RenderConfig &config, std::vector<float> &image, std::vector<int> &count,
                        const double eye[3], const double lookat[3], const double up[3], const double quat[4], int passes,
                        int step) {
  const int width = config.width, height = config.height;
  if (width <= 0 || height <= 0 || passes < 1 || step < 1) return false;
  if (image.size() < (size_t)3 * width * height || count.size() < (size_t)width * height) {
    printf("Mallie:err\tmsg:Render: image/count buffers are smaller than %dx%d\n", width, height);
    return false;
  }
  double origin[3], corner[3], du[3], dv[3];
  Camera camera(eye, lookat, up);
  camera.BuildCameraFrame(origin, corner, du, dv, config.fov, quat, width, height);

  if (gInitialPass) { // plane fixed by the FIRST scene rendered in this process (render.cc:615-628)
    gInitialPass = false;
    gPlane = config.plane;
    if (gPlane) {
      real3 bmin, bmax;
      scene.BoundingBox(bmin, bmax);
      float pl[4];
      plane_from_bbox(&bmin.x, &bmax.x, pl);
      gPlaneObject.set(pl[0], pl[1], pl[2], pl[3]);
    }
  }
  MgpuScene *dev = scene.DeviceScene();
  if (!dev) {
    printf("Mallie:err\tmsg:Render: no device scene (%s)\n", mgpu_last_error());
    return false;
  }
  {
    const char *e = getenv("MALLIE_FAST");
    const bool fast = gFastMode >= 0 ? gFastMode == 1 : (e && atoi(e) != 0);
    if (mgpu_scene_set_precision(dev, fast ? MGPU_PRECISION_FP32 : MGPU_PRECISION_FP64) != MGPU_OK) {
      printf("Mallie:err\tmsg:Render: %s\n", mgpu_last_error());
      return false;
    }
  }
  const float pl[4] = {gPlaneObject.m_a, gPlaneObject.m_b, gPlaneObject.m_c, gPlaneObject.m_d};
  const bool table = gRngTable != NULL;
  if (table && passes != 1) { // SetRenderRngTable holds the start states of ONE pass (width * height * 4 words)
    gRngTable = NULL;
    printf("Mallie:err\tmsg:RenderPasses: a start-state table covers one pass, %d were asked for\n", passes);
    return false;
  }
  MgpuStats st;
  memset(&st, 0, sizeof(st));
  const char *env_stream = getenv("MALLIE_RNG_STREAM");
  if ((gReferenceStream || (env_stream && atoi(env_stream) != 0)) && step == 1 && !table) {
    // the reference's own random stream (one OpenMP thread), continued from call to call like its static state
    const int rc = mgpu_render_stream(dev, origin, corner, du, dv, width, height, gMaxPathLength, passes, gPlane ? pl : NULL,
                                      gStreamState, &image[0], &count[0], NULL, &st);
    if (rc != MGPU_OK) {
      printf("Mallie:err\tmsg:Render failed: %s\n", mgpu_last_error());
      return false;
    }
    const double sec = st.total_ms / 1000.0;
    printf("\r[Mallie] Render time: %f sec(s) | %f fps", sec, sec > 0 ? 1.0 / sec : 0.0);
    fflush(stdout);
    return true;
  }
  const int gpus = (step == 1 && !table) ? wanted_gpus() : 1;
  const char *force_frame = getenv("MGPU_FRAME_FORCE_EXCHANGE"); // one GPU through the multi-GPU machinery (tests)
  if (gpus > 1 || (step == 1 && !table && force_frame && atoi(force_frame) != 0)) { // the same frame from n GPUs: strips, one RCCL exchange, assembled on device 0 (include/mgpu.h, mgpu_frame_*)
    MgpuFrame *mf = multi_frame(scene, gpus, width, height);
    if (!mf) return false;
    {
      const char *e = getenv("MALLIE_FAST");
      const bool fast = gFastMode >= 0 ? gFastMode == 1 : (e && atoi(e) != 0);
      for (size_t i = 1; i < gMulti.scenes.size(); i++) // the replicas follow the first scene's arithmetic
        if (mgpu_scene_set_precision(gMulti.scenes[i], fast ? MGPU_PRECISION_FP32 : MGPU_PRECISION_FP64) != MGPU_OK) return false;
    }
    const double cam[12] = {origin[0], origin[1], origin[2], corner[0], corner[1], corner[2],
                            du[0],     du[1],     du[2],     dv[0],     dv[1],     dv[2]};
    int slot = 0;
    int rc = mgpu_frame_render(mf, cam, gMaxPathLength, passes, gPlane ? pl : NULL, MGPU_RNG_HASH, gSeed, gPassCounter, &slot);
    if (rc == MGPU_OK) rc = mgpu_frame_wait(mf, slot, &image[0], NULL);
    if (rc != MGPU_OK) {
      printf("Mallie:err\tmsg:multi-GPU Render failed: %s\n", mgpu_frame_last_error());
      return false;
    }
    for (size_t i = 0; i < (size_t)width * height; i++) count[i] += passes;
    gPassCounter += (unsigned int)passes;
    printf("\r[Mallie] Render on %d GPUs", gpus);
    fflush(stdout);
    return true;
  }
  // A progressive driver calls Render() pass after pass with the same camera (main_sdl.cc:689, main_console.cc): the call
  // enqueues the NEXT pass on the device before it copies this one out (include/mgpu.h, mgpu_scene_set_render_ahead;
  // MALLIE_RENDER_AHEAD=0: never), which is why it asks for no statistics (they would tie the call to its own launch)
  static const bool render_ahead = !(getenv("MALLIE_RENDER_AHEAD") && atoi(getenv("MALLIE_RENDER_AHEAD")) == 0);
  const bool ahead = render_ahead && step == 1 && !table;
  if (mgpu_scene_set_render_ahead(dev, ahead ? 1 : 0) != MGPU_OK) {
    printf("Mallie:err\tmsg:Render: %s\n", mgpu_last_error());
    return false;
  }
  const auto t_call = std::chrono::steady_clock::now();
  const int rc = step == 1
                     ? mgpu_render(dev, origin, corner, du, dv, width, height, 0, 0, width, height, gMaxPathLength, passes,
                                   gPlane ? pl : NULL, table ? MGPU_RNG_TABLE : MGPU_RNG_HASH, gRngTable, gSeed, gPassCounter,
                                   &image[0], &count[0], ahead ? NULL : &st)
                     : mgpu_render_step(dev, origin, corner, du, dv, width, height, step, gMaxPathLength, gPlane ? pl : NULL,
                                        table ? MGPU_RNG_TABLE : MGPU_RNG_HASH, gRngTable, gSeed, gPassCounter, &image[0],
                                        &count[0], &st);
  gRngTable = NULL;
  if (rc != MGPU_OK) {
    printf("Mallie:err\tmsg:Render failed: %s\n", mgpu_last_error());
    return false;
  }
  gPassCounter += (unsigned int)passes;
  const double sec = ahead ? std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call).count() : st.total_ms / 1000.0;
  printf("\r[Mallie] Render time: %f sec(s) | %f fps", sec, sec > 0 ? 1.0 / sec : 0.0);
  fflush(stdout);
  return true;
}

bool RenderPasses(Scene &scene, const RenderConfig &config, std::vector<float> &image, std::vector<int> &count,
                  const double eye[3], const double lookat[3], const double up[3], const double quat[4], int passes) {
  return render_impl(scene, config, image, count, eye, lookat, up, quat, passes, 1);
}

// render.cc:593-708.  step > 1 (progressive block fill, render.cc:684-696): one path per step x step block, count += 3 per
// pixel as the reference's fill loop does; sizes that are not multiples of the step are refused (the reference writes
// outside the image there).
void Render(Scene &scene, const RenderConfig &config, std::vector<float> &image, std::vector<int> &count,
            const double eye[3], const double lookat[3], const double up[3], const double quat[4], int step) {
  if (step < 1) {
    printf("Mallie:err\tmsg:Render: step = %d\n", step);
    return;
  }
  render_impl(scene, config, image, count, eye, lookat, up, quat, 1, step);
}

// RenderPanoramic, render.cc:710-763 (what main_console.cc:111 calls): 10 PathTraceEnv samples per pixel of an
// equirectangular frame, maxPathLength 16; overwrites image, count[px] += 10.  Seeding as for Render (see the header).
void RenderPanoramic(Scene &scene, const RenderConfig &config, std::vector<float> &image, std::vector<int> &count,
                     const double eye[3], const double lookat[3], const double up[3], const double quat[4], bool stereo) {
  const int width = config.width, height = config.height;
  if (width <= 0 || height <= 0) return;
  if (image.size() < (size_t)3 * width * height || count.size() < (size_t)width * height) {
    printf("Mallie:err\tmsg:RenderPanoramic: image/count buffers are smaller than %dx%d\n", width, height);
    return;
  }
  double origin[3], corner[3], du[3], dv[3];
  Camera camera(eye, lookat, up);
  camera.BuildCameraFrame(origin, corner, du, dv, config.fov, quat, width, height);
  MgpuScene *dev = scene.DeviceScene();
  if (!dev) {
    printf("Mallie:err\tmsg:RenderPanoramic: no device scene (%s)\n", mgpu_last_error());
    return;
  }
  const bool table = gRngTable != NULL;
  MgpuStats st;
  const int rc = mgpu_render_panoramic(dev, origin, width, height, 0, 0, width, height, /*kMaxPathLength*/ 16, /*samples*/ 10,
                                       stereo ? 1 : 0, table ? MGPU_RNG_TABLE : MGPU_RNG_HASH, gRngTable, gSeed,
                                       gPassCounter, &image[0], &count[0], &st);
  gRngTable = NULL;
  if (rc != MGPU_OK) {
    printf("Mallie:err\tmsg:RenderPanoramic failed: %s\n", mgpu_last_error());
    return;
  }
  gPassCounter += 1;
  const double sec = st.total_ms / 1000.0;
  printf("\r[Mallie] Render time: %f sec(s) | %f fps", sec, sec > 0 ? 1.0 / sec : 0.0);
  fflush(stdout);
}

} // namespace mallie

extern "C" void mgpu_plane_from_bbox(const double bmin[3], const double bmax[3], float plane[4]) {
  mallie::plane_from_bbox(bmin, bmax, plane);
}
