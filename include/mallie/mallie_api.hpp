// include/mallie/mallie_api.hpp -- Mallie-compatible C++ surface of the MI355X render hot path.
//
// A caller written against lighttransport/mallie's render.cc-era headers (common.h, intersection.h, mesh.h,
// material.h, bvh_accel.h, scene.h, camera.h, prim-plane.h, render.h) compiles against the forwarding headers next to
// this file and links libmallie_mgpu.so instead of Mallie's own objects.  Names, namespaces, signatures, POD layouts
// and the bool/printf error convention follow the reference (file:line given per declaration); the work behind
// Scene::Trace / BVHAccel::Traverse / mallie::Render runs on the GPU through the C ABI of include/mgpu.h.
//
// Deviations a maintainer must know about (also in INTEGRATION.md):
//  * mallie::Render seeds paths per (pixel, pass) by default (MGPU_RNG_HASH, seed settable with mallie::SetRenderSeed: the
//    frame then does not depend on GPU count or scheduling), or from a caller-supplied table of start states
//    (mallie::SetRenderRngTable).  mallie::SetRenderReferenceStream(true) (or MALLIE_RNG_STREAM=1) selects the
//    reference's own stream instead -- its single-thread xorshift128 state continued from pixel to pixel and call to call
//    (render.cc:116-168 with OMP_NUM_THREADS=1): the image then IS the reference's image.
//  * MALLIE_GPUS=n in the environment makes mallie::Render / RenderPasses use n GPUs of the node (scene replicated,
//    interleaved 8-row strips, one RCCL exchange per frame; mgpu_frame_* in include/mgpu.h).  The image does not depend on n.
//  * mallie::SetRenderFastMode(true) (or MALLIE_FAST=1) switches Render / RenderPasses to fp32 arithmetic: faster, close to
//    the reference's image (rms per-pixel L2 2e-5 on the Cornell frame) but not identical to it.  Off by default.
//  * kMaxPathLength (render.cc:52) is a run-time setting here: mallie::SetMaxPathLength (default 16 = reference).
//  * Render() with step > 1 (progressive block fill, render.cc:684-696) needs a frame whose sizes are multiples of the
//    step: for other sizes the reference writes outside the image, and this implementation reports an error instead.
#ifndef MALLIE_MI355X_API_HPP_
#define MALLIE_MI355X_API_HPP_

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <string>
#include <vector>

// ---- common.h:6-83 -------------------------------------------------------------------------------------------------
typedef double real;

struct real3 {
  real x, y, z;

  real3() {}
  real3(real a, real b, real c) : x(a), y(b), z(c) {}
  real3(real *p) : x(p[0]), y(p[1]), z(p[2]) {}

  real operator[](int i) const { return (&x)[i]; }
  real &operator[](int i) { return (&x)[i]; }

  real3 operator+(const real3 &o) const { return real3(x + o.x, y + o.y, z + o.z); }
  real3 operator-(const real3 &o) const { return real3(x - o.x, y - o.y, z - o.z); }
  real3 operator*(const real3 &o) const { return real3(x * o.x, y * o.y, z * o.z); }
  real3 operator/(const real3 &o) const { return real3(x / o.x, y / o.y, z / o.z); }
  real3 operator*(real f) const { return real3(x * f, y * f, z * f); }
  real3 &operator+=(const real3 &o) {
    x += o.x;
    y += o.y;
    z += o.z;
    return *this;
  }
  real3 neg() { return real3(-x, -y, -z); }
  real length() { return std::sqrt(x * x + y * y + z * z); }
  // scales by 1/length only when the length exceeds 1e-6 (common.h:48-56)
  void normalize() {
    const real len = length();
    if (std::fabs(len) > 1.0e-6) {
      const real inv = 1.0 / len;
      x *= inv;
      y *= inv;
      z *= inv;
    }
  }
};

inline real3 operator*(real f, const real3 &v) { return real3(v.x * f, v.y * f, v.z * f); }
inline real3 vcross(real3 a, real3 b) {
  return real3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline real vdot(real3 a, real3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct Ray { // 88 bytes; only org/dir are inputs (the reference leaves the rest uninitialised, camera.cc:235-239)
  real3 org;
  real3 dir;
  real3 invDir;
  int dirSign[3];
};

// ---- intersection.h:6-24 (184 bytes) --------------------------------------------------------------------------------
typedef struct {
  real t, u, v;
  unsigned int faceID;
  unsigned int materialID;
  unsigned int f0, f1, f2;
  real3 position;
  real3 geometricNormal;
  real3 normal;
  real3 tangent;
  real3 binormal;
  real texcoord[2];
} Intersection;

// ---- mesh.h:7-18 (80 bytes) ------------------------------------------------------------------------------------------
typedef struct {
  size_t numVertices;
  size_t numFaces;
  real *vertices;                  // xyz * numVertices
  real *facevarying_normals;       // xyz * 3 * numFaces, may be NULL
  real *facevarying_tangents;      // unused
  real *facevarying_binormals;     // unused
  real *facevarying_uvs;           // uv * 3 * numFaces, may be NULL
  real *facevarying_vertex_colors; // unused
  unsigned int *faces;             // 3 * numFaces
  unsigned int *materialIDs;       // numFaces
} Mesh;

// ---- material.h:6-24 (80 bytes) ---------------------------------------------------------------------------------------
struct Material {
  real3 diffuse, reflection, refraction;
  int id;
  Material() : diffuse(0.5, 0.5, 0.5), reflection(0.0, 0.0, 0.0), refraction(0.0, 0.0, 0.0), id(-1) {}
};

// ---- bvh_accel.h:10-86 -------------------------------------------------------------------------------------------------
class BVHNode { // 64 bytes
public:
  BVHNode() {}
  ~BVHNode() {}
  real bmin[3];
  real bmax[3];
  int flag; // 1 = leaf, 0 = branch
  int axis;
  unsigned int data[2]; // leaf: {count, first index}; branch: {child0, child1}
};

struct BVHBuildOptions {
  bool debugPrint;
  real costTaabb;
  int minLeafPrimitives;
  int maxTreeDepth;
  int binSize;
  BVHBuildOptions() : debugPrint(false), costTaabb(0.2), minLeafPrimitives(16), maxTreeDepth(256), binSize(64) {}
};

struct BVHBuildStatistics {
  int maxTreeDepth;
  int numLeafNodes;
  int numBranchNodes;
  BVHBuildStatistics() : maxTreeDepth(0), numLeafNodes(0), numBranchNodes(0) {}
};

struct MgpuScene; // C-ABI handle (include/mgpu.h)

class BVHAccel {
public:
  BVHAccel();
  ~BVHAccel();

  // Binned-SAH build that reproduces the reference tree node for node (bvh_accel.cc:321-482).  Meshes of >= 65536
  // triangles are built on the GPU when one is present (same bytes, ~30x faster); MALLIE_BVH_BUILD=host|device overrides.
  bool Build(const Mesh *mesh, const BVHBuildOptions &options);
  // Extension: the host (CPU) builder only, whatever the size.
  bool BuildOnHost(const Mesh *mesh, const BVHBuildOptions &options);
  BVHBuildStatistics GetStatistics() const { return stats_; }
  // Same binary layout as the reference: u64 numNodes, BVHNode[], u64 numIndices, u32[] (bvh_accel.cc:484-544).
  bool Dump(const char *filename);
  bool Load(const char *filename);
  // Closest hit of ONE ray, on the device (a batch of one; use TraverseBatch for throughput).
  bool Traverse(Intersection &isect, const Mesh *mesh, Ray &ray);
  // Extension: n rays in one launch. isects[i] / hits[i] as Traverse would give them.
  bool TraverseBatch(Intersection *isects, unsigned char *hits, const Mesh *mesh, const Ray *rays, size_t n);

  const std::vector<BVHNode> &GetNodes() const { return nodes_; }
  const std::vector<unsigned int> &GetIndices() const { return indices_; }

  // Extension: the device scene bound to (mesh, this tree); created on first use. `materials` may be NULL.
  MgpuScene *DeviceScene(const Mesh *mesh, const std::vector<Material> *materials = NULL);
  void ReleaseDevice();

private:
  BVHAccel(const BVHAccel &);
  BVHAccel &operator=(const BVHAccel &);
  BVHBuildOptions options_;
  std::vector<BVHNode> nodes_;
  std::vector<unsigned int> indices_;
  BVHBuildStatistics stats_;
  MgpuScene *device_;
  const Mesh *device_mesh_;
};

namespace mallie {

// ---- scene.h:42-77 -----------------------------------------------------------------------------------------------------
class Scene {
public:
  Scene();
  ~Scene();

  // Loads a mesh file, scales it and builds the BVH (scene.cc:66-251). Wavefront .obj, .eson and MagicaVoxel .vox (which
  // also fills the 256 palette materials) are read by this library's own readers.
  bool Init(const std::string &objFilename, const std::string &esonFilename, const std::string &magicaVoxelFilename,
            const std::string &materialFilename, double sceneScale = 1.0, bool sceneFit = false);
  // Extension: adopt caller-built arrays (copied) instead of reading a file; applies scale/fit like Init.
  bool InitFromArrays(const real *vertices, size_t numVertices, const unsigned int *faces, size_t numFaces,
                      const unsigned int *materialIDs, const real *facevarying_normals, const real *facevarying_uvs,
                      const std::vector<Material> &materials, double sceneScale = 1.0, bool sceneFit = false);

  bool Trace(Intersection &isect, Ray &ray);
  void BoundingBox(real3 &bmin, real3 &bmax);
  real3 GetBackgroundRadiance(real3 &dir);
  const Material &GetMaterial(int matID) const {
    static Material s_default;
    if ((size_t)matID < materials_.size()) return materials_[matID];
    return s_default;
  }

  // Extensions used by mallie::Render and tests.
  MgpuScene *DeviceScene() { return accel_.DeviceScene(&mesh_, &materials_); }
  // a further copy of the scene in the HBM of `device` (multi-GPU rendering, MALLIE_GPUS); the caller releases it with
  // mgpu_scene_destroy
  MgpuScene *CreateDeviceScene(int device);
  const Mesh &GetMesh() const { return mesh_; }
  BVHAccel &GetAccel() { return accel_; }

protected:
  void FreeMesh();
  bool Finish(double sceneScale, bool sceneFit);
  Mesh mesh_;
  std::vector<Material> materials_;
  BVHAccel accel_;
};

// ---- camera.h:9-46 -----------------------------------------------------------------------------------------------------
class Camera {
public:
  Camera(const double eye[3], const double lookat[3], const double up[3]);
  ~Camera() {}
  void BuildCameraFrame(double origin[3], double corner[3], double u[3], double v[3], double fov, const double quat[4],
                        int width, int height);
  Ray GenerateRay(double u, double v) const;
  // camera.cc:242-329: equirectangular directions of RenderPanoramic (host code; the device kernel evaluates the same
  // expressions per sample)
  Ray GenerateEnvRay(double u, double v) const;
  Ray GenerateStereoEnvRay(double u, double v) const;

  double eye_[3], up_[3], lookat_[3];
  double origin_[3], corner_[3], du_[3], dv_[3]; // world space
  double fov_;
  int height_, width_;
};

// ---- prim-plane.h:10-26 --------------------------------------------------------------------------------------------------
class Plane {
public:
  Plane() : m_a(0), m_b(1), m_c(0), m_d(0) {}
  void set(float a, float b, float c, float d) {
    m_a = a;
    m_b = b;
    m_c = c;
    m_d = d;
  }
  // prim-plane.cc:8-44 (host code, float core; the device renderer evaluates the same test per ray)
  bool intersect(Intersection *info, const Ray &ray);
  float m_a, m_b, m_c, m_d;
};

// ---- render.h:11-61 -------------------------------------------------------------------------------------------------------
struct RenderConfig {
  double fov;
  int width;
  int height;
  double eye[3];
  double lookat[3];
  double up[3];
  double quat[4];
  double scene_scale;
  bool scene_fit;
  bool plane;
  int num_passes;
  int num_photons;
  std::string obj_filename;
  std::string eson_filename;
  std::string magicavoxel_filename;
  std::string material_filename;
  RenderConfig();
};

// One pass: overwrites image (3*W*H, RGB, top row first) with this pass's radiance and does count[px]++ (render.cc:593-708).
void Render(Scene &scene, const RenderConfig &config, std::vector<float> &image, std::vector<int> &count,
            const double eye[3], const double lookat[3], const double up[3], const double quat[4], int step);

// One panoramic frame (render.h:56-60, render.cc:710-763): equirectangular rays, 10 PathTraceEnv samples per pixel with
// the reference's kMaxPathLength = 16 whatever SetMaxPathLength says; overwrites image, count[px] += 10.  `stereo`: left
// eye in the top half of the frame, right eye in the bottom half (camera.cc:259-329).  Seeded like Render; with
// SetRenderRngTable the W*H*4 words are per-PIXEL start states (the pixel's 10 samples share one stream).
void RenderPanoramic(Scene &scene, const RenderConfig &config, std::vector<float> &image, std::vector<int> &count,
                     const double eye[3], const double lookat[3], const double up[3], const double quat[4], bool stereo);

// Extensions (see the header comment).
void SetMaxPathLength(int maxPathLength);          // default 16 (render.cc:52)
void SetRenderSeed(unsigned long long seed);       // default 1; also restarts the pass counter
void SetRenderRngTable(const unsigned int *states); // W*H*4 words for the NEXT Render() call only; NULL clears
// true: Render / RenderPasses draw from the reference's own serial stream (render.cc:116-168, one OpenMP thread), starting at
// its seed and continuing from call to call; resets that stream to the seed.  false: back to per-(pixel, pass) seeding.
void SetRenderReferenceStream(bool on);
// true (or MALLIE_FAST=1 in the environment): Render / RenderPasses compute in float (mgpu_scene_set_precision,
// MGPU_PRECISION_FP32): the same algorithm and random stream, ~1.3x faster, NOT bit-identical to the reference -- a path
// whose ray passes within ~1e-6 of a silhouette may decide differently and continue as another sample.
void SetRenderFastMode(bool on);
// `passes` passes in one launch, accumulated on the device in pass order (== Render + AccumImage, main_sdl.cc:138-143);
// count[px] += passes.  Returns false (after printing a Mallie:err line) on failure.
bool RenderPasses(Scene &scene, const RenderConfig &config, std::vector<float> &image, std::vector<int> &count,
                  const double eye[3], const double lookat[3], const double up[3], const double quat[4], int passes);

} // namespace mallie

#endif // MALLIE_MI355X_API_HPP_
