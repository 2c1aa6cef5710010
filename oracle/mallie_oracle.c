/*
 * oracle/mallie_oracle.c -- TEST INFRASTRUCTURE ONLY (see mallie_oracle.h for the rules).
 *
 * CPU restatement, in plain C and fp64, of the algorithm of lighttransport/mallie's render hot path.
 * Each function names the reference location it follows (paths relative to /root/reference).  The
 * arithmetic keeps the reference's operation ORDER and its float/double mixing, because the goal is
 * bit-for-bit equality with the reference binary (g++ -O2, x86-64, no FMA); build with -ffp-contract=off.
 *
 * Parity status: PINNED against goldens generated from the unmodified reference (tests/test_oracle_golden.py).
 */
#include "mallie_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define MO_NO_MATERIAL 0xFFFFFFFFu

/* ------------------------------------------------------------------------------------------------ */
/* small vector helpers (common.h:9-76): every op is elementwise, left-to-right                      */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { double x, y, z; } v3;

static inline v3 v3_make(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_scale(v3 a, double f) { return v3_make(a.x * f, a.y * f, a.z * f); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
/* common.h:66-72 */
static inline v3 v3_cross(v3 a, v3 b) {
  return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* common.h:74-76 */
static inline double v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* real3::normalize, common.h:46-56: scales by 1/len only when len > 1e-6 */
static inline v3 v3_normalized(v3 a) {
  double len = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  if (fabs(len) > 1.0e-6) {
    double inv = 1.0 / len;
    a.x *= inv; a.y *= inv; a.z *= inv;
  }
  return a;
}
static inline double v3_get(v3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
static inline v3 v3_load(const double *p) { return v3_make(p[0], p[1], p[2]); }

/* ------------------------------------------------------------------------------------------------ */
/* RNG                                                                                               */
/* ------------------------------------------------------------------------------------------------ */
/* render.cc:137-168 (Marsaglia xorshift128) */
double mo_xorshift128(uint32_t st[4]) {
  uint32_t t = st[0] ^ (st[0] << 11);
  st[0] = st[1];
  st[1] = st[2];
  st[2] = st[3];
  st[3] = (st[3] ^ (st[3] >> 19)) ^ (t ^ (t >> 8));
  return st[3] * (1.0 / 4294967296.0);
}

static inline uint64_t splitmix64_mix(uint64_t x) {
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

/* Not in the reference (its only seeding is per OpenMP thread, render.cc:118-135): the per-(pixel,pass) seeding the
 * GPU path and this oracle share for large runs. */
void mo_hash_state(uint64_t seed, uint32_t pass, uint32_t pixel, uint32_t st[4]) {
  const uint64_t golden = 0x9E3779B97F4A7C15ULL;
  uint64_t ctr = seed * golden + (((uint64_t)pass << 32) | (uint64_t)pixel);
  uint64_t a = splitmix64_mix(ctr + golden);
  uint64_t b = splitmix64_mix(ctr + 2 * golden);
  st[0] = (uint32_t)a;
  st[1] = (uint32_t)(a >> 32);
  st[2] = (uint32_t)b;
  st[3] = (uint32_t)(b >> 32);
  if ((st[0] | st[1] | st[2] | st[3]) == 0) st[0] = 1;
}

/* ------------------------------------------------------------------------------------------------ */
/* BVH build (bvh_accel.cc:36-482)                                                                   */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  const double *verts;
  const uint32_t *faces;
  uint32_t *indices;
  mo_node *nodes;
  size_t nn, cap;
  double costTaabb;
  int minLeaf, maxDepth, binSize;
  int statDepth, statLeaves, statBranches;
  size_t *bins; /* scratch: 2 * 3 * binSize */
  int oom;
} bvh_builder;

static size_t bb_push(bvh_builder *b) {
  if (b->nn == b->cap) {
    size_t ncap = b->cap ? b->cap * 2 : 1024;
    mo_node *p = (mo_node *)realloc(b->nodes, ncap * sizeof(mo_node));
    if (!p) { b->oom = 1; return 0; }
    b->nodes = p;
    b->cap = ncap;
  }
  memset(&b->nodes[b->nn], 0, sizeof(mo_node));
  return b->nn++;
}

/* bvh_accel.cc:285-315: box of all vertices of faces indices[l..r), padded by 1024*DBL_EPSILON per vertex */
static void bb_bounds(const bvh_builder *b, size_t l, size_t r, double bmin[3], double bmax[3]) {
  const double pad = DBL_EPSILON * 1024;
  size_t first = b->indices[l];
  const double *p = &b->verts[3 * (size_t)b->faces[3 * first]];
  for (int k = 0; k < 3; k++) { bmin[k] = p[k] - pad; bmax[k] = p[k] + pad; }
  for (size_t i = l; i < r; i++) {
    size_t face = b->indices[i];
    for (int j = 0; j < 3; j++) {
      const double *q = &b->verts[3 * (size_t)b->faces[3 * face + j]];
      for (int k = 0; k < 3; k++) {
        double lo = q[k] - pad, hi = q[k] + pad;
        if (bmin[k] > lo) bmin[k] = lo;
        if (bmax[k] < hi) bmax[k] = hi;
      }
    }
  }
}

/* bvh_accel.cc:55-80 */
static void tri_bounds(const bvh_builder *b, uint32_t face, double lo[3], double hi[3]) {
  const double *p0 = &b->verts[3 * (size_t)b->faces[3 * (size_t)face + 0]];
  for (int k = 0; k < 3; k++) lo[k] = hi[k] = p0[k];
  for (int j = 1; j < 3; j++) {
    const double *p = &b->verts[3 * (size_t)b->faces[3 * (size_t)face + j]];
    for (int k = 0; k < 3; k++) {
      lo[k] = lo[k] < p[k] ? lo[k] : p[k]; /* std::min(a,b): b<a ? b : a -- same value either way */
      hi[k] = hi[k] < p[k] ? p[k] : hi[k];
    }
  }
}

/* bvh_accel.cc:82-142: histogram of triangle-box min / max cell per axis */
static void bb_fill_bins(bvh_builder *b, const double smin[3], const double smax[3], size_t l, size_t r) {
  const double eps = DBL_EPSILON * 1024;
  const int nb = b->binSize;
  const double nbins = (double)nb;
  double inv[3];
  for (int k = 0; k < 3; k++) {
    double size = smax[k] - smin[k];
    inv[k] = (size > eps) ? nbins / size : 0.0;
  }
  memset(b->bins, 0, sizeof(size_t) * 2 * 3 * (size_t)nb); /* a fresh zeroed BinBuffer per node (bvh_accel.cc:377) */
  for (size_t i = l; i < r; i++) {
    double lo[3], hi[3];
    tri_bounds(b, b->indices[i], lo, hi);
    for (int k = 0; k < 3; k++) {
      double qlo = (lo[k] - smin[k]) * inv[k];
      double qhi = (hi[k] - smin[k]) * inv[k];
      size_t ilo = (unsigned int)floor(qlo);
      size_t ihi = (unsigned int)floor(qhi);
      if ((double)ilo >= nbins) ilo = (size_t)(nbins - 1);
      if ((double)ihi >= nbins) ihi = (size_t)(nbins - 1);
      b->bins[0 * (3 * nb) + k * nb + ilo] += 1;
      b->bins[1 * (3 * nb) + k * nb + ihi] += 1;
    }
  }
}

/* bvh_accel.cc:50-53 */
static inline double box_area(const double lo[3], const double hi[3]) {
  double bx = hi[0] - lo[0], by = hi[1] - lo[1], bz = hi[2] - lo[2];
  return 2.0 * (bx * by + by * bz + bz * bx);
}

/* bvh_accel.cc:144-255: sweep the 63 interior cell boundaries per axis, keep the cheapest cut per axis, then pick the
 * axis (strict '>' comparisons: earlier axis wins ties). */
static int bb_find_cut(const bvh_builder *b, const double bmin[3], const double bmax[3], size_t ntri, double cut[3]) {
  const double eps = DBL_EPSILON * 1024;
  const int nb = b->binSize;
  const double Taabb = b->costTaabb;
  const double Ttri = 1.0 - Taabb;
  double step[3];
  for (int k = 0; k < 3; k++) step[k] = (bmax[k] - bmin[k]) * (1.0 / nb);
  double total = box_area(bmin, bmax);
  double invTotal = (total > eps) ? 1.0 / total : 0.0;
  double best[3];
  for (int j = 0; j < 3; j++) {
    double bestPos = bmin[j] + 0.5 * step[j];
    best[j] = DBL_MAX;
    size_t left = 0, right = ntri;
    double loL[3], hiL[3], loR[3], hiR[3];
    for (int k = 0; k < 3; k++) { loL[k] = loR[k] = bmin[k]; hiL[k] = hiR[k] = bmax[k]; }
    for (int i = 0; i < nb - 1; i++) {
      left += b->bins[0 * (3 * nb) + j * nb + i];
      right -= b->bins[1 * (3 * nb) + j * nb + i];
      double pos = bmin[j] + (i + 0.5) * step[j];
      hiL[j] = pos;
      loR[j] = pos;
      double saL = box_area(loL, hiL);
      double saR = box_area(loR, hiR);
      /* SAH(), bvh_accel.cc:144-154 */
      double cost = 2.0 * Taabb + (saL * invTotal) * (double)left * Ttri + (saR * invTotal) * (double)right * Ttri;
      if (cost < best[j]) { best[j] = cost; bestPos = pos; }
    }
    cut[j] = bestPos;
  }
  int axis = 0;
  double c = best[0];
  if (c > best[1]) { axis = 1; c = best[1]; }
  if (c > best[2]) { axis = 2; c = best[2]; }
  return axis;
}

/* SAHPred, bvh_accel.cc:257-283 */
static inline int bb_goes_left(const bvh_builder *b, uint32_t face, int axis, double pos) {
  const uint32_t *f = &b->faces[3 * (size_t)face];
  double c = b->verts[3 * (size_t)f[0] + axis] + b->verts[3 * (size_t)f[1] + axis] + b->verts[3 * (size_t)f[2] + axis];
  return c < pos * 3.0;
}

/* The element order std::partition leaves matters for the resulting tree; this is libstdc++'s bidirectional
 * algorithm (GCC 11 bits/stl_algo.h, std::__partition(..., bidirectional_iterator_tag)), the one selected for the
 * raw pointers of bvh_accel.cc:391-402: advance `first` over accepted items, retreat `last` over rejected ones, swap. */
static size_t bb_partition(bvh_builder *b, size_t first, size_t last, int axis, double pos) {
  uint32_t *a = b->indices;
  for (;;) {
    for (;;) {
      if (first == last) return first;
      if (bb_goes_left(b, a[first], axis, pos)) ++first; else break;
    }
    --last;
    for (;;) {
      if (first == last) return first;
      if (!bb_goes_left(b, a[last], axis, pos)) --last; else break;
    }
    uint32_t tmp = a[first]; a[first] = a[last]; a[last] = tmp;
    ++first;
  }
}

/* BVHAccel::BuildTree, bvh_accel.cc:321-443 (depth-first, the branch slot is reserved before its subtrees) */
static size_t bb_build(bvh_builder *b, size_t l, size_t r, int depth) {
  size_t self = b->nn;
  if (b->statDepth < depth) b->statDepth = depth;
  double bmin[3], bmax[3];
  bb_bounds(b, l, r, bmin, bmax);
  size_t n = r - l;
  if (n < (size_t)b->minLeaf || depth >= b->maxDepth) {
    size_t id = bb_push(b);
    if (b->oom) return 0;
    mo_node *leaf = &b->nodes[id];
    memcpy(leaf->bmin, bmin, sizeof(bmin));
    memcpy(leaf->bmax, bmax, sizeof(bmax));
    leaf->flag = 1;
    leaf->axis = 0;
    leaf->data[0] = (uint32_t)n;
    leaf->data[1] = (uint32_t)l;
    b->statLeaves++;
    return self;
  }
  double cut[3] = {0.0, 0.0, 0.0};
  bb_fill_bins(b, bmin, bmax, l, r);
  int axis = bb_find_cut(b, bmin, bmax, n, cut);
  /* bvh_accel.cc:389-418: ONE axis is tried; a degenerate partition falls back to the object median (indices stay in
   * the order the partition left them). */
  size_t mid = bb_partition(b, l, r, axis, cut[axis]);
  if (mid == l || mid == r) mid = l + (n >> 1);
  size_t id = bb_push(b);
  if (b->oom) return 0;
  b->nodes[id].flag = 0;
  b->nodes[id].axis = axis;
  size_t c0 = bb_build(b, l, mid, depth + 1);
  if (b->oom) return 0;
  size_t c1 = bb_build(b, mid, r, depth + 1);
  if (b->oom) return 0;
  mo_node *nd = &b->nodes[id];
  nd->data[0] = (uint32_t)c0;
  nd->data[1] = (uint32_t)c1;
  memcpy(nd->bmin, bmin, sizeof(bmin));
  memcpy(nd->bmax, bmax, sizeof(bmax));
  b->statBranches++;
  return self;
}

int mo_bvh_build(const double *verts, size_t nv, const uint32_t *faces, size_t nf, double costTaabb,
                 int minLeafPrimitives, int maxTreeDepth, int binSize, mo_node **nodes_out, size_t *nn_out,
                 uint32_t **indices_out, int stats[3]) {
  (void)nv;
  if (!verts || !faces || nf == 0 || binSize < 2) return -1;
  bvh_builder b;
  memset(&b, 0, sizeof(b));
  b.verts = verts;
  b.faces = faces;
  b.costTaabb = costTaabb;
  b.minLeaf = minLeafPrimitives;
  b.maxDepth = maxTreeDepth;
  b.binSize = binSize;
  b.indices = (uint32_t *)malloc(sizeof(uint32_t) * nf);
  b.bins = (size_t *)malloc(sizeof(size_t) * 2 * 3 * (size_t)binSize);
  if (!b.indices || !b.bins) { free(b.indices); free(b.bins); return -2; }
  for (size_t i = 0; i < nf; i++) b.indices[i] = (uint32_t)i; /* bvh_accel.cc:460-463 */
  bb_build(&b, 0, nf, 0);
  free(b.bins);
  if (b.oom) { free(b.indices); free(b.nodes); return -2; }
  *nodes_out = b.nodes;
  *nn_out = b.nn;
  *indices_out = b.indices;
  if (stats) { stats[0] = b.statDepth; stats[1] = b.statLeaves; stats[2] = b.statBranches; }
  return 0;
}

void mo_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------------ */
/* scene                                                                                             */
/* ------------------------------------------------------------------------------------------------ */
struct mo_scene {
  size_t nv, nf, nn, nm;
  double *verts;
  uint32_t *faces;
  uint32_t *matIDs;   /* NULL allowed */
  double *fv_normals; /* NULL allowed */
  double *fv_uvs;     /* NULL allowed */
  mo_node *nodes;
  uint32_t *indices;
  double *mat_diffuse;
};

static void *dup_mem(const void *p, size_t bytes) {
  if (!p || !bytes) return NULL;
  void *q = malloc(bytes);
  if (q) memcpy(q, p, bytes);
  return q;
}

mo_scene *mo_scene_create(const double *verts, size_t nv, const uint32_t *faces, size_t nf, const uint32_t *matIDs,
                          const double *fv_normals, const double *fv_uvs, const mo_node *nodes, size_t nn,
                          const uint32_t *indices, const double *mat_diffuse, size_t nm) {
  if (!verts || !faces || !nodes || !indices || !nv || !nf || !nn) return NULL;
  mo_scene *s = (mo_scene *)calloc(1, sizeof(mo_scene));
  if (!s) return NULL;
  s->nv = nv; s->nf = nf; s->nn = nn; s->nm = nm;
  s->verts = (double *)dup_mem(verts, sizeof(double) * 3 * nv);
  s->faces = (uint32_t *)dup_mem(faces, sizeof(uint32_t) * 3 * nf);
  s->matIDs = (uint32_t *)dup_mem(matIDs, sizeof(uint32_t) * nf);
  s->fv_normals = (double *)dup_mem(fv_normals, sizeof(double) * 9 * nf);
  s->fv_uvs = (double *)dup_mem(fv_uvs, sizeof(double) * 6 * nf);
  s->nodes = (mo_node *)dup_mem(nodes, sizeof(mo_node) * nn);
  s->indices = (uint32_t *)dup_mem(indices, sizeof(uint32_t) * nf);
  s->mat_diffuse = (double *)dup_mem(mat_diffuse, sizeof(double) * 3 * nm);
  if (!s->verts || !s->faces || !s->nodes || !s->indices) { mo_scene_destroy(s); return NULL; }
  return s;
}

void mo_scene_destroy(mo_scene *s) {
  if (!s) return;
  free(s->verts); free(s->faces); free(s->matIDs); free(s->fv_normals); free(s->fv_uvs);
  free(s->nodes); free(s->indices); free(s->mat_diffuse);
  free(s);
}

/* scene.cc:317-333 */
void mo_scene_bbox(const mo_scene *s, double bmin[3], double bmax[3]) {
  for (int k = 0; k < 3; k++) { bmin[k] = s->nodes[0].bmin[k]; bmax[k] = s->nodes[0].bmax[k]; }
}

/* render.cc:620-627: float zmin, float zsize; plane = (0,1,0, -(zmin - zsize*0.0001f)) */
void mo_plane_from_bbox(const double bmin[3], const double bmax[3], float plane[4]) {
  float zmin = (float)bmin[1];
  float zsize = (float)(bmax[1] - bmin[1]);
  plane[0] = 0.0f;
  plane[1] = 1.0f;
  plane[2] = 0.0f;
  plane[3] = -(zmin - zsize * 0.0001f);
}

/* ------------------------------------------------------------------------------------------------ */
/* traversal (bvh_accel.cc:546-844)                                                                  */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  double t, u, v;
  uint32_t faceID, materialID;
  uint32_t f0, f1, f2;
  v3 position, geometricNormal, normal;
  double texcoord[2];
} isect_t;

#define MO_STACK 512 /* bvh_accel.cc:548,777 */

/* IntersectRayAABB, bvh_accel.cc:550-593 */
static inline int ray_box(const mo_node *nd, double maxT, v3 org, v3 inv, const int sign[3]) {
  double nx = sign[0] ? nd->bmax[0] : nd->bmin[0];
  double ny = sign[1] ? nd->bmax[1] : nd->bmin[1];
  double nz = sign[2] ? nd->bmax[2] : nd->bmin[2];
  double fx = sign[0] ? nd->bmin[0] : nd->bmax[0];
  double fy = sign[1] ? nd->bmin[1] : nd->bmax[1];
  double fz = sign[2] ? nd->bmin[2] : nd->bmax[2];
  double tmin_x = (nx - org.x) * inv.x, tmax_x = (fx - org.x) * inv.x;
  double tmin_y = (ny - org.y) * inv.y, tmax_y = (fy - org.y) * inv.y;
  double tmin = (tmin_x > tmin_y) ? tmin_x : tmin_y;
  double tmax = (tmax_x < tmax_y) ? tmax_x : tmax_y;
  double tmin_z = (nz - org.z) * inv.z, tmax_z = (fz - org.z) * inv.z;
  tmin = (tmin > tmin_z) ? tmin : tmin_z;
  tmax = (tmax < tmax_z) ? tmax : tmax_z;
  return (tmax > 0.0) && (tmin <= tmax) && (tmin <= maxT);
}

/* TriangleIsect, bvh_accel.cc:595-638 (Moller-Trumbore, no cull) */
static inline int ray_tri(double *tBest, double *uOut, double *vOut, v3 p0, v3 p1, v3 p2, v3 org, v3 dir) {
  const double eps = DBL_EPSILON * 1024;
  v3 e1 = v3_sub(p1, p0);
  v3 e2 = v3_sub(p2, p0);
  v3 p = v3_cross(dir, e2);
  double det = v3_dot(e1, p);
  if (fabs(det) < eps) return 0;
  double invDet = 1.0 / det;
  v3 s = v3_sub(org, p0);
  v3 q = v3_cross(s, e1);
  double u = v3_dot(s, p) * invDet;
  double v = v3_dot(q, dir) * invDet;
  double t = v3_dot(e2, q) * invDet;
  if (u < 0.0 || u > 1.0) return 0;
  if (v < 0.0 || u + v > 1.0) return 0;
  if (t < 0.0 || t > *tBest) return 0;
  *tBest = t;
  *uOut = u;
  *vOut = v;
  return 1;
}

/* BuildIntersection, bvh_accel.cc:699-769 */
static void finish_hit(const mo_scene *s, isect_t *is, v3 org, v3 dir) {
  const uint32_t *f = &s->faces[3 * (size_t)is->faceID];
  is->f0 = f[0]; is->f1 = f[1]; is->f2 = f[2];
  v3 p0 = v3_load(&s->verts[3 * (size_t)f[0]]);
  v3 p1 = v3_load(&s->verts[3 * (size_t)f[1]]);
  v3 p2 = v3_load(&s->verts[3 * (size_t)f[2]]);
  is->position = v3_make(org.x + is->t * dir.x, org.y + is->t * dir.y, org.z + is->t * dir.z);
  v3 n = v3_normalized(v3_cross(v3_sub(p1, p0), v3_sub(p2, p0)));
  is->geometricNormal = n;
  if (s->fv_normals) {
    const double *nn = &s->fv_normals[9 * (size_t)is->faceID];
    double w = 1.0 - is->u - is->v;
    is->normal.x = w * nn[0] + is->u * nn[3] + is->v * nn[6];
    is->normal.y = w * nn[1] + is->u * nn[4] + is->v * nn[7];
    is->normal.z = w * nn[2] + is->u * nn[5] + is->v * nn[8];
  } else {
    is->normal = n;
  }
  if (s->fv_uvs) {
    const double *uv = &s->fv_uvs[6 * (size_t)is->faceID];
    double w = 1.0 - is->u - is->v;
    is->texcoord[0] = w * uv[0] + is->u * uv[2] + is->v * uv[4];
    is->texcoord[1] = w * uv[1] + is->u * uv[3] + is->v * uv[5];
  }
}

/* BVHAccel::Traverse, bvh_accel.cc:773-844.  *nodes / *tris count popped nodes and triangle tests; *depth the
 * deepest stack index + 1. */
static int traverse(const mo_scene *s, isect_t *is, v3 org, v3 dir, uint64_t *nodes, uint64_t *tris, uint64_t *depth) {
  double hitT = DBL_MAX;
  int stack[MO_STACK];
  int sp = 0;
  stack[0] = 0;
  is->t = hitT;
  is->u = 0.0;
  is->v = 0.0;
  is->faceID = (uint32_t)-1;
  int sign[3] = {dir.x < 0.0 ? 1 : 0, dir.y < 0.0 ? 1 : 0, dir.z < 0.0 ? 1 : 0};
  v3 inv = v3_make(1.0 / dir.x, 1.0 / dir.y, 1.0 / dir.z);
  uint64_t nvis = 0, ntri = 0, dmax = 1;
  while (sp >= 0) {
    const mo_node *nd = &s->nodes[stack[sp]];
    sp--;
    nvis++;
    int hit = ray_box(nd, hitT, org, inv, sign);
    if (nd->flag == 0) {
      if (hit) {
        int nearIdx = sign[nd->axis];
        if (sp + 2 >= MO_STACK) return -1; /* the reference would overrun its 512-entry vector */
        stack[++sp] = (int)nd->data[1 - nearIdx];
        stack[++sp] = (int)nd->data[nearIdx];
        if ((uint64_t)sp + 1 > dmax) dmax = (uint64_t)sp + 1;
      }
    } else if (hit) {
      /* TestLeafNode, bvh_accel.cc:640-697 */
      uint32_t cnt = nd->data[0], first = nd->data[1];
      double t = is->t;
      int any = 0;
      for (uint32_t i = 0; i < cnt; i++) {
        uint32_t face = s->indices[first + i];
        const uint32_t *f = &s->faces[3 * (size_t)face];
        double u, v;
        ntri++;
        if (ray_tri(&t, &u, &v, v3_load(&s->verts[3 * (size_t)f[0]]), v3_load(&s->verts[3 * (size_t)f[1]]),
                    v3_load(&s->verts[3 * (size_t)f[2]]), org, dir)) {
          is->t = t;
          is->u = u;
          is->v = v;
          is->faceID = face;
          is->materialID = s->matIDs ? s->matIDs[face] : MO_NO_MATERIAL;
          any = 1;
        }
      }
      if (any) hitT = is->t;
    }
  }
  if (nodes) *nodes += nvis;
  if (tris) *tris += ntri;
  if (depth && dmax > *depth) *depth = dmax;
  if (is->t < DBL_MAX) {
    finish_hit(s, is, org, dir);
    return 1;
  }
  return 0;
}

int mo_trace(const mo_scene *s, const double *rays, size_t n, mo_hit *out, mo_stats *stats) {
  if (!s || (!rays && n) || (!out && n)) return -1;
  uint64_t nodes = 0, tris = 0, depth = 0;
  for (size_t i = 0; i < n; i++) {
    isect_t is;
    memset(&is, 0, sizeof(is));
    v3 org = v3_load(&rays[6 * i]), dir = v3_load(&rays[6 * i + 3]);
    int hit = traverse(s, &is, org, dir, &nodes, &tris, &depth);
    if (hit < 0) return -3;
    mo_hit *o = &out[i];
    memset(o, 0, sizeof(*o));
    if (hit) {
      o->hit = 1;
      o->faceID = is.faceID; o->materialID = is.materialID;
      o->f0 = is.f0; o->f1 = is.f1; o->f2 = is.f2;
      o->t = is.t; o->u = is.u; o->v = is.v;
      o->position[0] = is.position.x; o->position[1] = is.position.y; o->position[2] = is.position.z;
      o->geometricNormal[0] = is.geometricNormal.x; o->geometricNormal[1] = is.geometricNormal.y;
      o->geometricNormal[2] = is.geometricNormal.z;
      o->normal[0] = is.normal.x; o->normal[1] = is.normal.y; o->normal[2] = is.normal.z;
      o->texcoord[0] = is.texcoord[0]; o->texcoord[1] = is.texcoord[1];
    }
  }
  if (stats) {
    stats->trace_calls += n; stats->real_rays += n; stats->nodes += nodes; stats->tris += tris;
    if (depth > stats->max_stack) stats->max_stack = depth;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* plane (prim-plane.cc:8-44): float core, double outputs                                            */
/* ------------------------------------------------------------------------------------------------ */
static int plane_hit(const float pl[4], isect_t *is, v3 org, v3 dir) {
  v3 n = v3_make((double)pl[0], (double)pl[1], (double)pl[2]);
  v3 v = v3_normalized(dir);
  float vn = (float)v3_dot(v, n);
  if (fabsf(vn) > FLT_EPSILON * 1024.0f) {
    float on_d = (float)(v3_dot(org, n) + (double)pl[3]);
    float t = -on_d / vn;
    if ((t > 0) && ((double)t < is->t)) {
      is->t = (double)t;
      is->position = v3_add(org, v3_scale(v, (double)t));
      n = v3_normalized(n);
      is->geometricNormal = n;
      is->normal = n;
      is->texcoord[0] = 0.0;
      is->texcoord[1] = 0.0;
      is->materialID = MO_NO_MATERIAL;
      return 1;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* camera (camera.cc:12-240, matrix.cc:42-216, trackball.cc:268-291)                                 */
/* ------------------------------------------------------------------------------------------------ */
static double len3_guarded(const double v[3]) { /* vlength, camera.cc:22-28 / matrix.cc:18-24 */
  double l2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  return (fabs(l2) > 1.0e-30) ? sqrt(l2) : 0.0;
}
static void cross3(double c[3], const double a[3], const double b[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
/* matrix.cc:26-34 keeps the length in double ... */
static void normalize3_d(double v[3]) {
  double len = len3_guarded(v);
  if (fabs(len) > 1.0e-30) {
    double inv = 1.0 / len;
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
  }
}
/* ... camera.cc:30-38 rounds it to float first. */
static void normalize3_f(double v[3]) {
  float len = (float)len3_guarded(v);
  if (fabsf(len) > 1.0e-30) {
    double inv = 1.0 / len;
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
  }
}

/* Matrix::LookAt, matrix.cc:42-100 */
static void look_at(double m[4][4], const double eye[3], const double at[3], const double up[3]) {
  double look[3] = {at[0] - eye[0], at[1] - eye[1], at[2] - eye[2]};
  double u[3], v[3];
  normalize3_d(look);
  cross3(u, look, up);
  normalize3_d(u);
  cross3(v, u, look);
  normalize3_d(v);
  for (int k = 0; k < 3; k++) {
    m[0][k] = u[k];
    m[1][k] = v[k];
    m[2][k] = -look[k];
    m[3][k] = eye[k];
  }
  m[0][3] = m[1][3] = m[2][3] = 0.0;
  m[3][3] = 1.0;
}

/* Matrix::Inverse, matrix.cc:102-196: Cramer's rule over the transposed source.  Each cofactor is
 * (sum of three pair*elem products) - (sum of three pair*elem products); the tables give, per output entry, the
 * (pair index, source index) of the six products in the reference's evaluation order. */
static void inverse4(double m[4][4]) {
  double t[16], pr[12];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) t[i + 4 * j] = m[i][j];
  static const unsigned char pairA[12][2] = {{10, 15}, {11, 14}, {9, 15}, {11, 13}, {9, 14}, {10, 13},
                                             {8, 15},  {11, 12}, {8, 14}, {10, 12}, {8, 13}, {9, 12}};
  static const unsigned char pairB[12][2] = {{2, 7}, {3, 6}, {1, 7}, {3, 5}, {1, 6}, {2, 5},
                                             {0, 7}, {3, 4}, {0, 6}, {2, 4}, {0, 5}, {1, 4}};
  /* {pair,src} x3 added, then {pair,src} x3 subtracted */
  static const unsigned char cofA[8][12] = {
      {0, 5, 3, 6, 4, 7, 1, 5, 2, 6, 5, 7},   {1, 4, 6, 6, 9, 7, 0, 4, 7, 6, 8, 7},
      {2, 4, 7, 5, 10, 7, 3, 4, 6, 5, 11, 7}, {5, 4, 8, 5, 11, 6, 4, 4, 9, 5, 10, 6},
      {1, 1, 2, 2, 5, 3, 0, 1, 3, 2, 4, 3},   {0, 0, 7, 2, 8, 3, 1, 0, 6, 2, 9, 3},
      {3, 0, 6, 1, 11, 3, 2, 0, 7, 1, 10, 3}, {4, 0, 9, 1, 10, 2, 5, 0, 8, 1, 11, 2}};
  static const unsigned char cofB[8][12] = {
      {0, 13, 3, 14, 4, 15, 1, 13, 2, 14, 5, 15},   {1, 12, 6, 14, 9, 15, 0, 12, 7, 14, 8, 15},
      {2, 12, 7, 13, 10, 15, 3, 12, 6, 13, 11, 15}, {5, 12, 8, 13, 11, 14, 4, 12, 9, 13, 10, 14},
      {2, 10, 5, 11, 1, 9, 4, 11, 0, 9, 3, 10},     {8, 11, 0, 8, 7, 10, 6, 10, 9, 11, 1, 8},
      {6, 9, 11, 11, 3, 8, 10, 11, 2, 8, 7, 9},     {10, 10, 4, 8, 9, 9, 8, 9, 11, 0, 5, 8}};
  for (int half = 0; half < 2; half++) {
    const unsigned char(*pp)[2] = half ? pairB : pairA;
    const unsigned char(*cf)[12] = half ? cofB : cofA;
    for (int i = 0; i < 12; i++) pr[i] = t[pp[i][0]] * t[pp[i][1]];
    for (int e = 0; e < 8; e++) {
      const unsigned char *c = cf[e];
      double plus = pr[c[0]] * t[c[1]] + pr[c[2]] * t[c[3]] + pr[c[4]] * t[c[5]];
      double minus = pr[c[6]] * t[c[7]] + pr[c[8]] * t[c[9]] + pr[c[10]] * t[c[11]];
      double *dst = &m[half * 2 + e / 4][e % 4];
      *dst = plus;
      *dst -= minus;
    }
  }
  double det = t[0] * m[0][0] + t[1] * m[0][1] + t[2] * m[0][2] + t[3] * m[0][3];
  det = 1.0 / det;
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) m[j][i] *= det;
}

/* Matrix::Mult, matrix.cc:198-207: dst[i][j] = sum_k m0[k][j] * m1[i][k], accumulated from 0 */
static void mult4(double dst[4][4], double m0[4][4], double m1[4][4]) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double acc = 0;
      for (int k = 0; k < 4; k++) acc += m0[k][j] * m1[i][k];
      dst[i][j] = acc;
    }
}

/* Matrix::MultV, matrix.cc:209-216 */
static void multv(double dst[3], double m[4][4], const double v[3]) {
  for (int k = 0; k < 3; k++) dst[k] = m[0][k] * v[0] + m[1][k] * v[1] + m[2][k] * v[2] + m[3][k];
}

/* build_rotmatrix, trackball.cc:272-291 */
static void quat_matrix(double m[4][4], const double q[4]) {
  m[0][0] = 1.0 - 2.0 * (q[1] * q[1] + q[2] * q[2]);
  m[0][1] = 2.0 * (q[0] * q[1] - q[2] * q[3]);
  m[0][2] = 2.0 * (q[2] * q[0] + q[1] * q[3]);
  m[1][0] = 2.0 * (q[0] * q[1] + q[2] * q[3]);
  m[1][1] = 1.0 - 2.0 * (q[2] * q[2] + q[0] * q[0]);
  m[1][2] = 2.0 * (q[1] * q[2] - q[0] * q[3]);
  m[2][0] = 2.0 * (q[2] * q[0] - q[1] * q[3]);
  m[2][1] = 2.0 * (q[1] * q[2] + q[0] * q[3]);
  m[2][2] = 1.0 - 2.0 * (q[1] * q[1] + q[0] * q[0]);
  m[0][3] = m[1][3] = m[2][3] = 0.0;
  m[3][0] = m[3][1] = m[3][2] = 0.0;
  m[3][3] = 1.0;
}

/* Camera::BuildCameraFrame, camera.cc:40-220 */
void mo_camera_frame(const double eye[3], const double lookat[3], const double up[3], const double quat[4], double fov,
                     int width, int height, double frame[12]) {
  double r[4][4], re[4][4], m[4][4];
  quat_matrix(r, quat);
  double lo[3] = {lookat[0] - eye[0], lookat[1] - eye[1], lookat[2] - eye[2]};
  double dist = len3_guarded(lo);
  double dir[3] = {0.0, 0.0, dist};
  inverse4(r);
  const double zero[3] = {0.0, 0.0, 0.0};
  const double localUp[3] = {0.0, 1.0, 0.0};
  look_at(re, dir, zero, localUp);
  re[3][0] += eye[0];
  re[3][1] += eye[1];
  re[3][2] += (eye[2] - dist);
  mult4(m, r, re);
  double eye1[3], lookat1[3];
  multv(eye1, m, zero);
  dir[2] = -dir[2];
  multv(lookat1, m, dir);
  /* camera.cc:141-144: the caller's up vector is used as is */
  double up1[3] = {up[0], up[1], up[2]};
  /* camera.cc:174-199 */
  double flen = (0.5f * (double)height / tanf(0.5f * (double)(fov * M_PI / 180.0f)));
  double look1[3] = {lookat1[0] - eye1[0], lookat1[1] - eye1[1], lookat1[2] - eye1[2]};
  double *origin = &frame[0], *corner = &frame[3], *u = &frame[6], *v = &frame[9];
  cross3(u, look1, up1);
  normalize3_f(u);
  cross3(v, look1, u);
  normalize3_f(v);
  normalize3_f(look1);
  for (int k = 0; k < 3; k++) look1[k] = flen * look1[k] + eye1[k];
  for (int k = 0; k < 3; k++) corner[k] = look1[k] - 0.5f * (width * u[k] + height * v[k]);
  for (int k = 0; k < 3; k++) origin[k] = eye1[k];
}

/* Camera::GenerateRay, camera.cc:222-240 */
void mo_generate_ray(const double frame[12], double u, double v, double ray[6]) {
  const double *origin = &frame[0], *corner = &frame[3], *du = &frame[6], *dv = &frame[9];
  v3 d;
  d.x = (corner[0] + u * du[0] + v * dv[0]) - origin[0];
  d.y = (corner[1] + u * du[1] + v * dv[1]) - origin[1];
  d.z = (corner[2] + u * du[2] + v * dv[2]) - origin[2];
  d = v3_normalized(d);
  ray[0] = origin[0]; ray[1] = origin[1]; ray[2] = origin[2];
  ray[3] = d.x; ray[4] = d.y; ray[5] = d.z;
}

/* ------------------------------------------------------------------------------------------------ */
/* path tracing (render.cc:271-456)                                                                  */
/* ------------------------------------------------------------------------------------------------ */
/* GenerateBasis, render.cc:271-317: minor axis by |n[i]| compared AFTER rounding to float (fabsf) */
static void make_basis(v3 *tangent, v3 *binormal, v3 n) {
  int index = -1;
  double minval = 1.0e+6;
  for (int i = 0; i < 3; i++) {
    double val = (double)fabsf((float)v3_get(n, i));
    if (val < minval) { minval = val; index = i; }
  }
  v3 t;
  if (index == 0) t = v3_make(0.0, -n.z, n.y);
  else if (index == 1) t = v3_make(-n.z, 0.0, n.x);
  else t = v3_make(-n.y, n.x, 0.0);
  t = v3_normalized(t);
  *tangent = t;
  *binormal = v3_normalized(v3_cross(t, n));
}

/* SampleDiffuseIS, render.cc:320-339 */
static v3 sample_diffuse(v3 n, uint32_t rng[4]) {
  v3 tangent, binormal;
  make_basis(&tangent, &binormal, n);
  double theta = acos(sqrt(1.0 - mo_xorshift128(rng)));
  double phi = 2.0 * M_PI * mo_xorshift128(rng);
  double cos_theta = cos(theta);
  v3 T = v3_scale(v3_scale(tangent, cos(phi)), sin(theta));
  v3 B = v3_scale(v3_scale(binormal, sin(phi)), sin(theta));
  v3 N = v3_scale(n, cos_theta);
  return v3_add(v3_add(T, B), N);
}

typedef struct {
  uint64_t trace_calls, real_rays, nodes, tris, garbage_nodes, garbage_hits, paths, max_stack;
} path_counters;

/* PathTrace, render.cc:381-456.  Nothing is short-circuited: after the first miss the path keeps iterating with the
 * stale intersection record exactly as the reference does; those Trace() calls are only COUNTED apart. */
static int path_trace(const mo_scene *s, const double frame[12], const float *plane, int maxPathLength, int px, int py,
                      uint32_t rng[4], double radiance_out[3], path_counters *pc, double *probe, int *probe_n) {
  float ju = (float)(mo_xorshift128(rng) - 0.5);
  float jv = (float)(mo_xorshift128(rng) - 0.5);
  double ray[6];
  mo_generate_ray(frame, (double)((float)px + ju), (double)((float)py + jv), ray);
  v3 org = v3_load(&ray[0]), dir = v3_load(&ray[3]);
  isect_t is;
  memset(&is, 0, sizeof(is));
  is.t = 1.0e+30;
  double thr[3] = {1.0, 1.0, 1.0};
  double rad[3] = {0.0, 0.0, 0.0};
  int escaped = 0;
  pc->paths++;
  for (unsigned pathLength = 1;; ++pathLength) {
    uint64_t nv = 0, nt = 0;
    int hit = traverse(s, &is, org, dir, &nv, &nt, &pc->max_stack);
    if (hit < 0) return -3;
    pc->trace_calls++;
    if (escaped) {
      pc->garbage_nodes += nv;
      if (hit) pc->garbage_hits++;
    } else {
      pc->real_rays++;
      pc->nodes += nv;
      pc->tris += nt;
    }
    int mesh_final = hit; /* the mesh hit survives unless the plane is closer */
    if (plane) {
      int ph = plane_hit(plane, &is, org, dir);
      if (ph && escaped) pc->garbage_hits++;
      if (ph) mesh_final = 0;
      hit |= ph;
    }
    if (probe && !escaped) { /* same record as mgpu_probe_path (include/mgpu.h) */
      double *rec = probe + (size_t)(pathLength - 1) * 16;
      rec[0] = org.x; rec[1] = org.y; rec[2] = org.z; rec[3] = dir.x; rec[4] = dir.y; rec[5] = dir.z;
      rec[6] = is.t; rec[7] = hit ? 1.0 : 0.0;
      rec[8] = mesh_final ? (double)is.faceID : -1.0;
      rec[9] = is.normal.x; rec[10] = is.normal.y; rec[11] = is.normal.z; rec[12] = (double)is.materialID;
      rec[13] = (double)pathLength; rec[14] = thr[0]; rec[15] = rad[0];
      *probe_n = (int)pathLength;
    }
    if (!hit) {
      if (pathLength < 2) break; /* kMinPathLength */
      escaped = 1;
      double L = (double)pathLength;
      rad[0] += thr[0] * 0.5 / L;
      rad[1] += thr[1] * 0.5 / L;
      rad[2] += thr[2] * 0.5 / L;
    }
    if (pathLength >= (unsigned)maxPathLength) break;
    v3 hitP = v3_add(org, v3_scale(dir, is.t));
    (void)mo_xorshift128(rng); /* `double r = randomreal();` -- drawn, never used (render.cc:430) */
    v3 n = is.normal;
    double ndoti = v3_dot(is.normal, v3_neg(dir));
    if (ndoti < 0.0) n = v3_neg(n);
    v3 sd = sample_diffuse(n, rng);
    if (is.materialID != MO_NO_MATERIAL) {
      /* Scene::GetMaterial(int), scene.h:58-65: (size_t)(int)id < materials_.size() ? materials_[id] : default 0.5 */
      size_t id = (size_t)(int)is.materialID;
      if (id < s->nm) {
        thr[0] *= s->mat_diffuse[3 * id + 0];
        thr[1] *= s->mat_diffuse[3 * id + 1];
        thr[2] *= s->mat_diffuse[3 * id + 2];
      } else {
        thr[0] *= 0.5; thr[1] *= 0.5; thr[2] *= 0.5;
      }
    }
    org = v3_add(hitP, v3_scale(sd, 1.0e-3));
    dir = sd;
    is.t = 1.0e+30;
  }
  radiance_out[0] = rad[0]; radiance_out[1] = rad[1]; radiance_out[2] = rad[2];
  return 0;
}

static void merge_stats(mo_stats *dst, const path_counters *pc) {
  dst->trace_calls += pc->trace_calls;
  dst->real_rays += pc->real_rays;
  dst->nodes += pc->nodes;
  dst->tris += pc->tris;
  dst->garbage_nodes += pc->garbage_nodes;
  dst->garbage_hits += pc->garbage_hits;
  dst->paths += pc->paths;
  if (pc->max_stack > dst->max_stack) dst->max_stack = pc->max_stack;
}

int mo_render(const mo_scene *s, const double frame[12], int W, int H, int x0, int y0, int x1, int y1,
              int maxPathLength, int passes, const float *plane, int rng_mode, uint32_t stream_state[4],
              const uint32_t *rng_states, uint64_t seed, uint32_t pass_base, float *image, int32_t *count,
              uint32_t *states_out, mo_stats *stats, int nthreads) {
  if (!s || !frame || !image || W <= 0 || H <= 0 || passes < 1 || maxPathLength < 1) return -1;
  if (x0 < 0 || y0 < 0 || x1 > W || y1 > H || x0 > x1 || y0 > y1) return -1;
  if (rng_mode == MO_RNG_STREAM && !stream_state) return -1;
  if (rng_mode == MO_RNG_TABLE && !rng_states) return -1;
  if (rng_mode < 0 || rng_mode > 2) return -1;
  int err = 0;
  path_counters total;
  memset(&total, 0, sizeof(total));

  if (rng_mode == MO_RNG_STREAM) {
    /* scanline order, one state, as the reference with OMP_NUM_THREADS=1 (render.cc:657-681) */
    for (int p = 0; p < passes; p++) {
      for (int y = y0; y < y1; y++) {
        for (int x = x0; x < x1; x++) {
          size_t px = (size_t)y * W + x;
          if (states_out) memcpy(&states_out[((size_t)p * W * H + px) * 4], stream_state, 16);
          double rad[3];
          if (path_trace(s, frame, plane, maxPathLength, x, y, stream_state, rad, &total, NULL, NULL)) return -3;
          for (int c = 0; c < 3; c++) {
            float f = (float)rad[c];
            image[3 * px + c] = (p == 0) ? f : image[3 * px + c] + f;
          }
        }
      }
    }
  } else {
#ifdef _OPENMP
    if (nthreads < 1) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
      path_counters local;
      memset(&local, 0, sizeof(local));
#pragma omp for schedule(dynamic, 1)
      for (int y = y0; y < y1; y++) {
        for (int x = x0; x < x1; x++) {
          size_t px = (size_t)y * W + x;
          float acc[3] = {0.0f, 0.0f, 0.0f};
          for (int p = 0; p < passes; p++) {
            uint32_t st[4];
            if (rng_mode == MO_RNG_TABLE) memcpy(st, &rng_states[((size_t)p * W * H + px) * 4], 16);
            else mo_hash_state(seed, pass_base + (uint32_t)p, (uint32_t)px, st);
            if (states_out) memcpy(&states_out[((size_t)p * W * H + px) * 4], st, 16);
            double rad[3];
            if (path_trace(s, frame, plane, maxPathLength, x, y, st, rad, &local, NULL, NULL)) {
#pragma omp atomic write
              err = -3;
            }
            for (int c = 0; c < 3; c++) {
              float f = (float)rad[c];
              acc[c] = (p == 0) ? f : acc[c] + f;
            }
          }
          image[3 * px + 0] = acc[0];
          image[3 * px + 1] = acc[1];
          image[3 * px + 2] = acc[2];
        }
      }
#pragma omp critical
      {
        total.trace_calls += local.trace_calls; total.real_rays += local.real_rays;
        total.nodes += local.nodes; total.tris += local.tris;
        total.garbage_nodes += local.garbage_nodes; total.garbage_hits += local.garbage_hits;
        total.paths += local.paths;
        if (local.max_stack > total.max_stack) total.max_stack = local.max_stack;
      }
    }
  }
  if (count) {
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) count[(size_t)y * W + x] += passes;
  }
  if (stats) merge_stats(stats, &total);
  return err;
}

/* ShowNormal (mode 0, render.cc:458-485) / ShowUV (mode 1, render.cc:487-516) for every pixel, in scanline order: two
 * draws of jitter, the primary ray, Scene::Trace (mesh only, no plane), colour from the hit's shading normal
 * (n * 0.5 + 0.5) or its texture coordinate (0.1 * s, 0, 0); black on a miss.  A mesh without facevarying_uvs leaves
 * Intersection::texcoord uninitialised in the reference; it reads as 0 here.  RNG modes as mo_render (one pass). */
int mo_render_aov(const mo_scene *s, const double frame[12], int W, int H, int mode, int rng_mode, uint32_t stream_state[4],
                  const uint32_t *rng_states, uint64_t seed, uint32_t pass_base, float *image, uint32_t *states_out,
                  mo_stats *stats) {
  if (!s || !frame || !image || W <= 0 || H <= 0 || mode < 0 || mode > 1) return -1;
  if (rng_mode == MO_RNG_STREAM && !stream_state) return -1;
  if (rng_mode == MO_RNG_TABLE && !rng_states) return -1;
  uint64_t nodes = 0, tris = 0, depth = 0, rays = 0;
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      size_t px = (size_t)y * W + x;
      uint32_t st[4], *rng = st;
      if (rng_mode == MO_RNG_STREAM) rng = stream_state;
      else if (rng_mode == MO_RNG_TABLE) memcpy(st, &rng_states[px * 4], 16);
      else mo_hash_state(seed, pass_base, (uint32_t)px, st);
      if (states_out) memcpy(&states_out[px * 4], rng, 16);
      float ju = (float)(mo_xorshift128(rng) - 0.5);
      float jv = (float)(mo_xorshift128(rng) - 0.5);
      double ray[6];
      mo_generate_ray(frame, (double)((float)x + ju), (double)((float)y + jv), ray);
      isect_t is;
      memset(&is, 0, sizeof(is));
      int hit = traverse(s, &is, v3_load(&ray[0]), v3_load(&ray[3]), &nodes, &tris, &depth);
      if (hit < 0) return -3;
      rays++;
      double rad[3] = {0.0, 0.0, 0.0};
      if (hit) {
        if (mode == 0) {
          rad[0] = is.normal.x * 0.5 + 0.5;
          rad[1] = is.normal.y * 0.5 + 0.5;
          rad[2] = is.normal.z * 0.5 + 0.5;
        } else {
          rad[0] = 0.1 * is.texcoord[0];
        }
      }
      for (int c = 0; c < 3; c++) image[3 * px + c] = (float)rad[c];
    }
  if (stats) {
    stats->trace_calls += rays; stats->real_rays += rays; stats->nodes += nodes; stats->tris += tris; stats->paths += rays;
    if (depth > stats->max_stack) stats->max_stack = depth;
  }
  return 0;
}

/* One Render() call with its `step` argument (render.cc:657-696): one path per step x step block -- the block's top-left
 * pixel -- in scanline order, then the block fill, which increments count once per colour channel (3 per pixel); with
 * step == 1 count is incremented once (render.cc:677-679).  W and H must be multiples of step (the reference's fill
 * writes outside the image otherwise).  RNG modes as mo_render, start states indexed by the block's top-left pixel. */
int mo_render_step(const mo_scene *s, const double frame[12], int W, int H, int step, int maxPathLength,
                   const float *plane, int rng_mode, uint32_t stream_state[4], const uint32_t *rng_states, uint64_t seed,
                   uint32_t pass_base, float *image, int32_t *count, uint32_t *states_out, mo_stats *stats) {
  if (!s || !frame || !image || W <= 0 || H <= 0 || step < 1 || maxPathLength < 1) return -1;
  if (W % step || H % step) return -1;
  if (rng_mode == MO_RNG_STREAM && !stream_state) return -1;
  if (rng_mode == MO_RNG_TABLE && !rng_states) return -1;
  path_counters total;
  memset(&total, 0, sizeof(total));
  memset(image, 0, sizeof(float) * 3 * (size_t)W * H); /* render.cc:641 */
  for (int y = 0; y < H; y += step) {
    for (int x = 0; x < W; x += step) {
      size_t px = (size_t)y * W + x;
      uint32_t st[4], *rng = st;
      if (rng_mode == MO_RNG_STREAM) rng = stream_state;
      else if (rng_mode == MO_RNG_TABLE) memcpy(st, &rng_states[px * 4], 16);
      else mo_hash_state(seed, pass_base, (uint32_t)px, st);
      if (states_out) memcpy(&states_out[px * 4], rng, 16);
      double rad[3];
      if (path_trace(s, frame, plane, maxPathLength, x, y, rng, rad, &total, NULL, NULL)) return -3;
      for (int c = 0; c < 3; c++) image[3 * px + c] = (float)rad[c];
      if (step == 1 && count) count[px]++;
    }
    if (step > 1) {
      for (int x = 0; x < W; x += step)
        for (int v = 0; v < step; v++)
          for (int u = 0; u < step; u++)
            for (int k = 0; k < 3; k++) {
              image[((size_t)(y + v) * W * 3 + (size_t)(x + u) * 3) + k] = image[3 * ((size_t)y * W + x) + k];
              if (count) count[(size_t)(y + v) * W + (x + u)]++;
            }
    }
  }
  if (stats) merge_stats(stats, &total);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* RenderPanoramic (render.cc:710-763) = PathTraceEnv (render.cc:518-590) over equirectangular rays    */
/* ------------------------------------------------------------------------------------------------ */
/* Camera::GenerateEnvRay (camera.cc:242-257) / Camera::GenerateStereoEnvRay (camera.cc:259-329).  origin = the camera
 * frame's origin_ (BuildCameraFrame); W, H = the frame size.  Stereo: top half = left eye, bottom half = right eye,
 * eyes on a circle of radius 0.5 toed in towards a focal distance of 4. */
void mo_generate_env_ray(const double origin[3], int W, int H, int stereo, double u, double v, double ray[6]) {
  if (!stereo) {
    double theta = M_PI * (v / H);
    double phi = 2.0 * M_PI * (u / W);
    ray[0] = origin[0]; ray[1] = origin[1]; ray[2] = origin[2];
    ray[3] = sin(theta) * cos(phi);
    ray[4] = cos(theta);
    ray[5] = sin(theta) * sin(phi);
    return;
  }
  const int is_left_side = v < (double)(H >> 1);
  const double focal_length = 4.0;
  const double r = 0.5;
  double theta = M_PI * fmod(2.0 * v / H, 1.0);
  double phi = 2.0 * M_PI * (u / W);
  v3 d0 = v3_make(sin(theta) * cos(phi), cos(theta), sin(theta) * sin(phi));
  v3 parallax = is_left_side ? v3_make(-d0.z, 0.0, d0.x) : v3_make(d0.z, 0.0, -d0.x);
  parallax = v3_normalized(parallax);
  parallax = v3_scale(parallax, r);
  ray[0] = origin[0] + parallax.x;
  ray[1] = origin[1] + parallax.y;
  ray[2] = origin[2] + parallax.z;
  double psi = atan2(r, focal_length);
  if (is_left_side) psi = -psi;
  v3 d = v3_make(d0.x * cos(psi) - d0.z * sin(psi), d0.y, d0.x * sin(psi) + d0.z * cos(psi));
  d = v3_normalized(d);
  ray[3] = d.x; ray[4] = d.y; ray[5] = d.z;
}

/* PathTraceEnv, render.cc:518-590: no plane, no material, `throughput` never used; a miss at length L >= 2 adds
 * 0.5/L and -- as in PathTrace -- does NOT end the path: the loop runs on to maxPathLength with the stale
 * record, every later ray starting ~1e308 away. */
static int path_trace_env(const mo_scene *s, const double origin[3], int W, int H, int stereo, int maxPathLength, int px,
                          int py, uint32_t rng[4], double radiance_out[3], path_counters *pc) {
  float ju = (float)(mo_xorshift128(rng) - 0.5);
  float jv = (float)(mo_xorshift128(rng) - 0.5);
  double ray[6];
  mo_generate_env_ray(origin, W, H, stereo, (double)((float)px + ju), (double)((float)py + jv), ray);
  v3 org = v3_load(&ray[0]), dir = v3_load(&ray[3]);
  isect_t is;
  memset(&is, 0, sizeof(is));
  is.t = 1.0e+30;
  double rad[3] = {0.0, 0.0, 0.0};
  int escaped = 0;
  pc->paths++;
  for (unsigned pathLength = 1;; ++pathLength) {
    uint64_t nv = 0, nt = 0;
    int hit = traverse(s, &is, org, dir, &nv, &nt, &pc->max_stack);
    if (hit < 0) return -3;
    pc->trace_calls++;
    if (escaped) {
      pc->garbage_nodes += nv;
      if (hit) pc->garbage_hits++;
    } else {
      pc->real_rays++;
      pc->nodes += nv;
      pc->tris += nt;
    }
    if (!hit) {
      if (pathLength < 2) break; /* kMinPathLength */
      escaped = 1;
      double L = (double)pathLength; /* kd / real3(pathLength, pathLength, pathLength), kd = 0.5 */
      rad[0] += 0.5 / L;
      rad[1] += 0.5 / L;
      rad[2] += 0.5 / L;
    }
    if (pathLength >= (unsigned)maxPathLength) break;
    v3 hitP = v3_add(org, v3_scale(dir, is.t));
    (void)mo_xorshift128(rng); /* `double r = randomreal();` -- drawn, never used (render.cc:563) */
    v3 n = is.normal;
    double ndoti = v3_dot(is.normal, v3_neg(dir));
    if (ndoti < 0.0) n = v3_neg(n);
    v3 sd = sample_diffuse(n, rng);
    org = v3_add(hitP, v3_scale(sd, 1.0e-3));
    dir = sd;
    is.t = 1.0e+30;
  }
  radiance_out[0] = rad[0]; radiance_out[1] = rad[1]; radiance_out[2] = rad[2];
  return 0;
}

int mo_render_panoramic(const mo_scene *s, const double origin[3], int W, int H, int x0, int y0, int x1, int y1,
                        int maxPathLength, int samples, int stereo, int rng_mode, uint32_t stream_state[4],
                        const uint32_t *rng_states, uint64_t seed, uint32_t pass_base, float *image, int32_t *count,
                        uint32_t *states_out, mo_stats *stats, int nthreads) {
  if (!s || !origin || !image || W <= 0 || H <= 0 || samples < 1 || maxPathLength < 1) return -1;
  if (x0 < 0 || y0 < 0 || x1 > W || y1 > H || x0 > x1 || y0 > y1) return -1;
  if (rng_mode == MO_RNG_STREAM && !stream_state) return -1;
  if (rng_mode == MO_RNG_TABLE && !rng_states) return -1;
  if (rng_mode < 0 || rng_mode > 2) return -1;
  int err = 0;
  path_counters total;
  memset(&total, 0, sizeof(total));
#ifdef _OPENMP
  if (nthreads < 1) nthreads = omp_get_max_threads();
#else
  nthreads = 1;
#endif
  if (rng_mode == MO_RNG_STREAM) nthreads = 1; /* scanline order, one state (OMP_NUM_THREADS=1 in the reference) */
#pragma omp parallel num_threads(nthreads)
  {
    path_counters local;
    memset(&local, 0, sizeof(local));
#pragma omp for schedule(dynamic, 1)
    for (int y = y0; y < y1; y++) {
      for (int x = x0; x < x1; x++) {
        size_t px = (size_t)y * W + x;
        uint32_t own[4];
        uint32_t *st = own;
        if (rng_mode == MO_RNG_STREAM) st = stream_state;
        else if (rng_mode == MO_RNG_TABLE) memcpy(own, &rng_states[px * 4], 16);
        else mo_hash_state(seed, pass_base, (uint32_t)px, own);
        if (states_out) memcpy(&states_out[px * 4], st, 16);
        float acc[3] = {0.0f, 0.0f, 0.0f}; /* memset(image) then `image[...] += radiance[k]`: float += double */
        for (int i = 0; i < samples; i++) {
          double rad[3];
          if (path_trace_env(s, origin, W, H, stereo, maxPathLength, x, y, st, rad, &local)) {
#pragma omp atomic write
            err = -3;
          }
          for (int c = 0; c < 3; c++) acc[c] = (float)((double)acc[c] + rad[c]);
        }
        image[3 * px + 0] = acc[0];
        image[3 * px + 1] = acc[1];
        image[3 * px + 2] = acc[2];
        if (count) count[px] += samples;
      }
    }
#pragma omp critical
    {
      total.trace_calls += local.trace_calls; total.real_rays += local.real_rays;
      total.nodes += local.nodes; total.tris += local.tris;
      total.garbage_nodes += local.garbage_nodes; total.garbage_hits += local.garbage_hits;
      total.paths += local.paths;
      if (local.max_stack > total.max_stack) total.max_stack = local.max_stack;
    }
  }
  if (stats) merge_stats(stats, &total);
  return err;
}

/* One eye path with per-iteration records (16 doubles each, layout of mgpu_probe_path in include/mgpu.h, except that
 * field 8 holds the FACE id of a mesh hit rather than the BVH slot).  Only iterations up to and including the first
 * miss are recorded, which is what the device executes. */
int mo_probe_path(const mo_scene *s, const double frame[12], int px, int py, int maxPathLength, const float *plane,
                  const uint32_t start_state[4], double *records, int *n_records, double radiance[3]) {
  if (!s || !frame || !start_state || !records || !n_records) return -1;
  uint32_t st[4];
  memcpy(st, start_state, 16);
  path_counters pc;
  memset(&pc, 0, sizeof(pc));
  double rad[3];
  *n_records = 0;
  int rc = path_trace(s, frame, plane, maxPathLength, px, py, st, rad, &pc, records, n_records);
  if (radiance) { radiance[0] = rad[0]; radiance[1] = rad[1]; radiance[2] = rad[2]; }
  return rc;
}

/* ------------------------------------------------------------------------------------------------ */
/* display transforms of the two drivers (main_console.cc:25-43; main_sdl.cc:157-165,420-477)        */
/* ------------------------------------------------------------------------------------------------ */
static unsigned char to_byte(double scaled) {
  /* `int i = x * 255.5;` on x86-64 is cvttsd2si: NaN and out-of-range give INT_MIN */
  int i;
  if (!(scaled < 2147483648.0) || scaled < -2147483648.0) i = (-2147483647 - 1);
  else i = (int)scaled;
  return (unsigned char)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

void mo_tonemap(const float *image, const int32_t *count, size_t npix, int mode, unsigned char *out) {
  for (size_t px = 0; px < npix; px++) {
    const int c = count[px];
    if (mode == 0) { /* HDRToLDR: out[i] = fclamp(in[i] / in_count[i / 3]) */
      for (int k = 0; k < 3; k++) out[3 * px + k] = to_byte((double)(image[3 * px + k] / (float)c) * 255.5);
    } else { /* Display: scale = 1.0f / count; BGRA; fclamp with gamma 2.2 */
      const float scale = 1.0f / (float)c;
      const float gamma = 2.2f;
      out[4 * px + 2] = to_byte((double)powf(scale * image[3 * px + 0], 1.0f / gamma) * 255.5);
      out[4 * px + 1] = to_byte((double)powf(scale * image[3 * px + 1], 1.0f / gamma) * 255.5);
      out[4 * px + 0] = to_byte((double)powf(scale * image[3 * px + 2], 1.0f / gamma) * 255.5);
      out[4 * px + 3] = 255;
    }
  }
}
