// host/bvh_build.cc -- BVHAccel::Build / Dump / Load and the device binding of the Mallie-compatible facade.
//
// The builder is host code, like the reference's (bvh_accel.cc:36-482): a top-down binned-SAH build whose result must
// equal the reference tree node for node, because traversal order, exact-t tie breaks and the node/triangle visit
// counts quoted in DESIGN.md all depend on it.  What has to be matched:
//   * per-node box = union of vertex boxes, each padded by 1024*DBL_EPSILON                (bvh_accel.cc:285-315)
//   * 64 bins/axis; a triangle adds 1 to the cell of its box minimum and 1 to the cell of its maximum    (:82-142)
//   * candidate planes at bmin + (i+0.5)*step, i = 0..62; strict '<' keeps the first best; axis choice x,y,z by
//     strict '>'                                                                                         (:156-255)
//   * split by centroid*3 < plane*3 with libstdc++'s bidirectional std::partition element order        (:257-283,402)
//   * a degenerate split falls back to the object median; leaves hold < 16 triangles; depth <= 256      (:341,405-409)
// It is written as an explicit work stack (children pushed right-then-left) which yields the same pre-order numbering
// as the reference's recursion.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../../include/mallie/mallie_api.hpp"
#include "../../../include/mgpu.h"

namespace {

const double kPad = DBL_EPSILON * 1024;

struct Box {
  double lo[3], hi[3];
  double area() const {
    const double a = hi[0] - lo[0], b = hi[1] - lo[1], c = hi[2] - lo[2];
    return 2.0 * (a * b + b * c + c * a);
  }
};

class Builder {
public:
  Builder(const Mesh *m, const BVHBuildOptions &o, std::vector<BVHNode> &nodes, std::vector<unsigned int> &idx,
          BVHBuildStatistics &st)
      : v_(m->vertices), f_(m->faces), opt_(o), nodes_(nodes), idx_(idx), st_(st), hist_(6 * (size_t)o.binSize) {}

  void run(size_t nfaces) {
    struct Task { size_t l, r; int depth; long parent; int side; };
    std::vector<Task> work;
    work.push_back(Task{0, nfaces, 0, -1, 0});
    while (!work.empty()) {
      const Task t = work.back();
      work.pop_back();
      const size_t self = nodes_.size();
      if (t.parent >= 0) nodes_[(size_t)t.parent].data[t.side] = (unsigned int)self;
      if (st_.maxTreeDepth < t.depth) st_.maxTreeDepth = t.depth;
      Box box;
      bounds(t.l, t.r, box);
      const size_t n = t.r - t.l;
      BVHNode nd;
      memset((void *)&nd, 0, sizeof(nd));
      for (int k = 0; k < 3; k++) { nd.bmin[k] = box.lo[k]; nd.bmax[k] = box.hi[k]; }
      if (n < (size_t)opt_.minLeafPrimitives || t.depth >= opt_.maxTreeDepth) {
        nd.flag = 1;
        nd.data[0] = (unsigned int)n;
        nd.data[1] = (unsigned int)t.l;
        nodes_.push_back(nd);
        st_.numLeafNodes++;
        continue;
      }
      double cut[3];
      histogram(box, t.l, t.r);
      const int axis = choose_cut(box, n, cut);
      size_t mid = split(t.l, t.r, axis, cut[axis]);
      if (mid == t.l || mid == t.r) mid = t.l + (n >> 1);
      nd.flag = 0;
      nd.axis = axis;
      nodes_.push_back(nd);
      st_.numBranchNodes++;
      work.push_back(Task{mid, t.r, t.depth + 1, (long)self, 1});
      work.push_back(Task{t.l, mid, t.depth + 1, (long)self, 0});
    }
  }

private:
  const double *vert(unsigned int face, int corner) const { return v_ + 3 * (size_t)f_[3 * (size_t)face + corner]; }

  void bounds(size_t l, size_t r, Box &b) const {
    const double *p = vert(idx_[l], 0);
    for (int k = 0; k < 3; k++) { b.lo[k] = p[k] - kPad; b.hi[k] = p[k] + kPad; }
    for (size_t i = l; i < r; i++)
      for (int c = 0; c < 3; c++) {
        const double *q = vert(idx_[i], c);
        for (int k = 0; k < 3; k++) {
          const double lo = q[k] - kPad, hi = q[k] + kPad;
          if (b.lo[k] > lo) b.lo[k] = lo;
          if (b.hi[k] < hi) b.hi[k] = hi;
        }
      }
  }

  void histogram(const Box &b, size_t l, size_t r) {
    const int nb = opt_.binSize;
    const double fb = (double)nb;
    double scale[3];
    for (int k = 0; k < 3; k++) {
      const double ext = b.hi[k] - b.lo[k];
      scale[k] = (ext > kPad) ? fb / ext : 0.0;
    }
    std::fill(hist_.begin(), hist_.end(), (size_t)0);
    for (size_t i = l; i < r; i++) {
      const unsigned int face = idx_[i];
      for (int k = 0; k < 3; k++) {
        const double a = vert(face, 0)[k], c1 = vert(face, 1)[k], c2 = vert(face, 2)[k];
        const double tlo = std::min(std::min(a, c1), c2), thi = std::max(std::max(a, c1), c2);
        size_t ilo = (unsigned int)std::floor((tlo - b.lo[k]) * scale[k]);
        size_t ihi = (unsigned int)std::floor((thi - b.lo[k]) * scale[k]);
        if ((double)ilo >= fb) ilo = (size_t)(fb - 1);
        if ((double)ihi >= fb) ihi = (size_t)(fb - 1);
        hist_[(size_t)k * nb + ilo]++;                  // "min" histogram
        hist_[(size_t)(3 + k) * nb + ihi]++;            // "max" histogram
      }
    }
  }

  int choose_cut(const Box &b, size_t n, double cut[3]) const {
    const int nb = opt_.binSize;
    const double Ta = opt_.costTaabb, Tt = 1.0 - opt_.costTaabb;
    const double total = b.area();
    const double inv_total = (total > kPad) ? 1.0 / total : 0.0;
    double best_cost[3];
    for (int j = 0; j < 3; j++) {
      const double step = (b.hi[j] - b.lo[j]) * (1.0 / nb);
      double best_pos = b.lo[j] + 0.5 * step;
      best_cost[j] = DBL_MAX;
      Box L = b, R = b;
      size_t nl = 0, nr = n;
      for (int i = 0; i < nb - 1; i++) {
        nl += hist_[(size_t)j * nb + i];
        nr -= hist_[(size_t)(3 + j) * nb + i];
        const double pos = b.lo[j] + (i + 0.5) * step;
        L.hi[j] = pos;
        R.lo[j] = pos;
        const double cost = 2.0 * Ta + (L.area() * inv_total) * (double)nl * Tt + (R.area() * inv_total) * (double)nr * Tt;
        if (cost < best_cost[j]) { best_cost[j] = cost; best_pos = pos; }
      }
      cut[j] = best_pos;
    }
    int axis = 0;
    double c = best_cost[0];
    if (c > best_cost[1]) { axis = 1; c = best_cost[1]; }
    if (c > best_cost[2]) { axis = 2; }
    return axis;
  }

  bool left_of(unsigned int face, int axis, double pos) const {
    const double c = vert(face, 0)[axis] + vert(face, 1)[axis] + vert(face, 2)[axis];
    return c < pos * 3.0;
  }

  // std::partition is specified only up to the resulting grouping; the reference's tree depends on the element order
  // libstdc++ produces, so that two-pointer scheme is spelled out here instead of calling the library.
  size_t split(size_t first, size_t last, int axis, double pos) {
    for (;;) {
      while (first != last && left_of(idx_[first], axis, pos)) ++first;
      if (first == last) return first;
      --last;
      while (first != last && !left_of(idx_[last], axis, pos)) --last;
      if (first == last) return first;
      std::swap(idx_[first], idx_[last]);
      ++first;
    }
  }

  const double *v_;
  const unsigned int *f_;
  const BVHBuildOptions &opt_;
  std::vector<BVHNode> &nodes_;
  std::vector<unsigned int> &idx_;
  BVHBuildStatistics &st_;
  std::vector<size_t> hist_;
};

} // namespace

BVHAccel::BVHAccel() : device_(NULL), device_mesh_(NULL) {}
BVHAccel::~BVHAccel() { ReleaseDevice(); }

bool BVHAccel::BuildOnHost(const Mesh *mesh, const BVHBuildOptions &options) {
  if (!mesh || !mesh->vertices || !mesh->faces || options.binSize < 2) return false;
  ReleaseDevice();
  options_ = options;
  stats_ = BVHBuildStatistics();
  nodes_.clear();
  const size_t n = mesh->numFaces;
  indices_.resize(n);
  for (size_t i = 0; i < n; i++) indices_[i] = (unsigned int)i;
  if (n == 0) return true;
  Builder(mesh, options_, nodes_, indices_, stats_).run(n);
  return true;
}

bool BVHAccel::Build(const Mesh *mesh, const BVHBuildOptions &options) {
  if (!mesh || !mesh->vertices || !mesh->faces || options.binSize < 2) return false;
  ReleaseDevice();
  options_ = options;
  stats_ = BVHBuildStatistics();
  nodes_.clear();
  const size_t n = mesh->numFaces;
  indices_.resize(n);
  for (size_t i = 0; i < n; i++) indices_[i] = (unsigned int)i;
  if (n == 0) return true; // the reference leaves an empty tree for an empty mesh (bvh_accel.cc:470)
  // Large meshes are built on the GPU when one is present (mgpu_bvh_build_device: the same tree, byte for byte, ~15x
  // faster at 1M-10M triangles); MALLIE_BVH_BUILD=host|device overrides the size rule.
  const char *mode = getenv("MALLIE_BVH_BUILD");
  const bool want_device = mode ? (strcmp(mode, "device") == 0) : (n >= 65536 && options_.binSize <= 256);
  if (want_device && !(mode && strcmp(mode, "host") == 0) && mgpu_device_count() > 0) {
    int dev = 0;
    if (const char *e = getenv("MALLIE_DEVICE")) dev = atoi(e);
    else if (const char *e2 = getenv("LOCAL_RANK")) dev = atoi(e2) % std::max(1, mgpu_device_count());
    MgpuNode *dn = NULL;
    uint32_t *di = NULL;
    size_t nn = 0;
    int st[3] = {0, 0, 0};
    static_assert(sizeof(BVHNode) == sizeof(MgpuNode), "BVHNode must be the 64-byte reference layout");
    const int rc = mgpu_bvh_build_device(mesh->vertices, mesh->numVertices, mesh->faces, n, options_.costTaabb,
                                         options_.minLeafPrimitives, options_.maxTreeDepth, options_.binSize, dev, &dn, &nn,
                                         &di, st, NULL);
    if (rc == MGPU_OK) {
      nodes_.resize(nn);
      memcpy((void *)&nodes_[0], dn, sizeof(BVHNode) * nn);
      memcpy(&indices_[0], di, sizeof(unsigned int) * n);
      mgpu_free(dn);
      mgpu_free(di);
      stats_.maxTreeDepth = st[0];
      stats_.numLeafNodes = st[1];
      stats_.numBranchNodes = st[2];
      return true;
    }
    printf("Mallie:info\tmsg:device BVH build unavailable (%d), building on the host\n", rc);
  }
  Builder(mesh, options_, nodes_, indices_, stats_).run(n);
  return true;
}

bool BVHAccel::Dump(const char *filename) {
  FILE *fp = fopen(filename, "wb");
  if (!fp) {
    fprintf(stderr, "[BVHAccel] Cannot write a file: %s\n", filename);
    return false;
  }
  const unsigned long long nn = nodes_.size(), ni = indices_.size();
  bool ok = fwrite(&nn, sizeof(nn), 1, fp) == 1;
  ok = ok && (nn == 0 || fwrite(&nodes_[0], sizeof(BVHNode), nn, fp) == nn);
  ok = ok && fwrite(&ni, sizeof(ni), 1, fp) == 1;
  ok = ok && (ni == 0 || fwrite(&indices_[0], sizeof(unsigned int), ni, fp) == ni);
  fclose(fp);
  return ok;
}

bool BVHAccel::Load(const char *filename) {
  FILE *fp = fopen(filename, "rb");
  if (!fp) {
    fprintf(stderr, "Cannot open file: %s\n", filename);
    return false;
  }
  ReleaseDevice();
  unsigned long long nn = 0, ni = 0;
  bool ok = fread(&nn, sizeof(nn), 1, fp) == 1 && nn > 0 && nn < (1ull << 32);
  if (ok) {
    nodes_.resize(nn);
    ok = fread(&nodes_[0], sizeof(BVHNode), nn, fp) == nn;
  }
  ok = ok && fread(&ni, sizeof(ni), 1, fp) == 1 && ni < (1ull << 32);
  if (ok) {
    indices_.resize(ni);
    ok = ni == 0 || fread(&indices_[0], sizeof(unsigned int), ni, fp) == ni;
  }
  fclose(fp);
  return ok;
}

MgpuScene *BVHAccel::DeviceScene(const Mesh *mesh, const std::vector<Material> *materials) {
  if (device_ && device_mesh_ == mesh) return device_;
  ReleaseDevice();
  if (!mesh || nodes_.empty()) return NULL;
  std::vector<double> diffuse;
  if (materials)
    for (size_t i = 0; i < materials->size(); i++)
      for (int k = 0; k < 3; k++) diffuse.push_back((*materials)[i].diffuse[k]);
  static_assert(sizeof(BVHNode) == sizeof(MgpuNode), "BVHNode must be the 64-byte reference layout");
  int dev = 0;
  if (const char *e = getenv("MALLIE_DEVICE")) dev = atoi(e);
  else if (const char *e2 = getenv("LOCAL_RANK")) dev = atoi(e2) % std::max(1, mgpu_device_count());
  MgpuScene *s = NULL;
  const int rc = mgpu_scene_create(mesh->vertices, mesh->numVertices, mesh->faces, mesh->numFaces, mesh->materialIDs,
                                   mesh->facevarying_normals, mesh->facevarying_uvs,
                                   reinterpret_cast<const MgpuNode *>(&nodes_[0]), nodes_.size(), &indices_[0],
                                   diffuse.empty() ? NULL : &diffuse[0], diffuse.size() / 3, dev, &s);
  if (rc != MGPU_OK) {
    printf("Mallie:err\tmsg:GPU scene upload failed: %s\n", mgpu_last_error());
    return NULL;
  }
  device_ = s;
  device_mesh_ = mesh;
  return device_;
}

void BVHAccel::ReleaseDevice() {
  if (device_) mgpu_scene_destroy(device_);
  device_ = NULL;
  device_mesh_ = NULL;
}

bool BVHAccel::TraverseBatch(Intersection *isects, unsigned char *hits, const Mesh *mesh, const Ray *rays, size_t n) {
  static_assert(sizeof(Ray) == sizeof(MgpuRay) && sizeof(Intersection) == sizeof(MgpuIntersection), "POD layouts");
  MgpuScene *s = DeviceScene(mesh, NULL);
  if (!s) return false;
  const int rc = mgpu_trace(s, reinterpret_cast<const MgpuRay *>(rays), n, reinterpret_cast<MgpuIntersection *>(isects),
                            hits, NULL);
  if (rc != MGPU_OK) {
    printf("Mallie:err\tmsg:GPU trace failed: %s\n", mgpu_last_error());
    return false;
  }
  return true;
}

bool BVHAccel::Traverse(Intersection &isect, const Mesh *mesh, Ray &ray) {
  // Only the fields Traverse itself writes are taken from the device record; on a miss the caller's stale
  // position/normal/materialID survive, as in the reference (bvh_accel.cc:782-786).
  Intersection tmp;
  unsigned char hit = 0;
  if (!TraverseBatch(&tmp, &hit, mesh, &ray, 1)) return false;
  if (hit) {
    const real3 keepT = isect.tangent, keepB = isect.binormal;
    const real keepU = isect.texcoord[0], keepV = isect.texcoord[1];
    isect = tmp;
    isect.tangent = keepT; // the mesh path never writes tangent/binormal (bvh_accel.cc:699-769)
    isect.binormal = keepB;
    if (!mesh->facevarying_uvs) { isect.texcoord[0] = keepU; isect.texcoord[1] = keepV; }
    return true;
  }
  isect.t = tmp.t;
  isect.u = 0.0;
  isect.v = 0.0;
  isect.faceID = (unsigned int)-1;
  return false;
}

// ---- C-ABI host helpers (include/mgpu.h) ------------------------------------------------------------------------------
extern "C" int mgpu_bvh_build(const double *verts, size_t nv, const uint32_t *faces, size_t nf, double costTaabb,
                              int minLeafPrimitives, int maxTreeDepth, int binSize, MgpuNode **nodes_out,
                              size_t *nn_out, uint32_t **indices_out, int stats[3]) {
  if (!verts || !faces || !nodes_out || !nn_out || !indices_out || nf == 0 || nv == 0 || binSize < 2) return MGPU_ERR_INVALID;
  Mesh m;
  memset(&m, 0, sizeof(m));
  m.numVertices = nv;
  m.numFaces = nf;
  m.vertices = const_cast<double *>(verts);
  m.faces = const_cast<uint32_t *>(faces);
  BVHBuildOptions o;
  o.costTaabb = costTaabb;
  o.minLeafPrimitives = minLeafPrimitives;
  o.maxTreeDepth = maxTreeDepth;
  o.binSize = binSize;
  BVHAccel acc;
  if (!acc.BuildOnHost(&m, o)) return MGPU_ERR_INVALID;
  const std::vector<BVHNode> &n = acc.GetNodes();
  const std::vector<unsigned int> &ix = acc.GetIndices();
  MgpuNode *pn = (MgpuNode *)malloc(sizeof(MgpuNode) * n.size());
  uint32_t *pi = (uint32_t *)malloc(sizeof(uint32_t) * ix.size());
  if (!pn || !pi) {
    free(pn);
    free(pi);
    return MGPU_ERR_OOM;
  }
  memcpy(pn, &n[0], sizeof(MgpuNode) * n.size());
  memcpy(pi, &ix[0], sizeof(uint32_t) * ix.size());
  *nodes_out = pn;
  *nn_out = n.size();
  *indices_out = pi;
  if (stats) {
    const BVHBuildStatistics st = acc.GetStatistics();
    stats[0] = st.maxTreeDepth;
    stats[1] = st.numLeafNodes;
    stats[2] = st.numBranchNodes;
  }
  return MGPU_OK;
}

extern "C" void mgpu_free(void *p) { free(p); }
