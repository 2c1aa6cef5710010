/* tools/wide_proto.c -- CPU prototype of the "wide node" traversal order used by the gfx950 kernels:
 * an interior node's record holds BOTH child boxes, the near child is entered directly, the far child goes on the stack
 * with its exact tmin and is re-checked against the (possibly shrunk) best t when popped.  This program only answers two
 * design questions on the CPU -- (1) is the result and the node count identical to the reference order (pop, test, push
 * far, push near; bvh_accel.cc:805-834)?  (2) how many far entries does a ray ever hold? -- it is not part of the product
 * and not an oracle.  Build: gcc -O2 -ffp-contract=off -shared -fPIC -o tools/libwide_proto.so tools/wide_proto.c
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
  double bmin[3], bmax[3];
  int32_t flag, axis;
  uint32_t data[2];
} node_t;

typedef struct {
  double t, u, v;
  uint32_t slot;
  uint32_t nodes, tris, max_sp, pops_failed;
} res_t;

static int slab(const double *bmin, const double *bmax, const double *o, const double *inv, const int *sg, double bt,
                double *tmin_out) {
  /* IntersectRayAABB, literal form (bvh_accel.cc:550-593) */
  const double nx = sg[0] ? bmax[0] : bmin[0], fx = sg[0] ? bmin[0] : bmax[0];
  const double ny = sg[1] ? bmax[1] : bmin[1], fy = sg[1] ? bmin[1] : bmax[1];
  const double nz = sg[2] ? bmax[2] : bmin[2], fz = sg[2] ? bmin[2] : bmax[2];
  const double tminx = (nx - o[0]) * inv[0], tmaxx = (fx - o[0]) * inv[0];
  const double tminy = (ny - o[1]) * inv[1], tmaxy = (fy - o[1]) * inv[1];
  double tmin = (tminx > tminy) ? tminx : tminy;
  double tmax = (tmaxx < tmaxy) ? tmaxx : tmaxy;
  const double tminz = (nz - o[2]) * inv[2], tmaxz = (fz - o[2]) * inv[2];
  tmin = (tmin > tminz) ? tmin : tminz;
  tmax = (tmax < tmaxz) ? tmax : tmaxz;
  *tmin_out = tmin;
  return (tmax > 0.0) && (tmin <= tmax) && (tmin <= bt);
}

/* tris: slot order, 9 doubles each: p0, e1, e2 */
static void leaf(const double *tris, uint32_t first, uint32_t cnt, const double *o, const double *d, res_t *r) {
  for (uint32_t i = 0; i < cnt; i++) {
    const double *T = tris + 9 * (size_t)(first + i);
    const double *p0 = T, *e1 = T + 3, *e2 = T + 6;
    r->tris++;
    double p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
    double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (fabs(det) < DBL_EPSILON * 1024) continue;
    double invDet = 1.0 / det;
    double s[3] = {o[0] - p0[0], o[1] - p0[1], o[2] - p0[2]};
    double q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
    double u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * invDet;
    double v = (q[0] * d[0] + q[1] * d[1] + q[2] * d[2]) * invDet;
    double t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * invDet;
    if (u < 0.0 || u > 1.0) continue;
    if (v < 0.0 || u + v > 1.0) continue;
    if (t < 0.0 || t > r->t) continue;
    r->t = t; r->u = u; r->v = v; r->slot = first + i;
  }
}

/* reference order */
void proto_ref(const node_t *nodes, const double *tris, const double *rays, size_t n, res_t *out) {
  for (size_t k = 0; k < n; k++) {
    const double *o = rays + 6 * k, *d = o + 3;
    double inv[3] = {1.0 / d[0], 1.0 / d[1], 1.0 / d[2]};
    int sg[3] = {d[0] < 0.0, d[1] < 0.0, d[2] < 0.0};
    res_t r = {DBL_MAX, 0, 0, 0xffffffffu, 0, 0, 0, 0};
    uint32_t st[512];
    int sp = 0;
    st[0] = 0;
    while (sp >= 0) {
      const node_t *nd = nodes + st[sp--];
      double tm;
      r.nodes++;
      int hit = slab(nd->bmin, nd->bmax, o, inv, sg, r.t, &tm);
      if (nd->flag == 0) {
        if (hit) {
          int o1 = sg[nd->axis], o0 = 1 - o1;
          st[++sp] = nd->data[o1];
          st[++sp] = nd->data[o0];
          if ((uint32_t)(sp + 1) > r.max_sp) r.max_sp = sp + 1;
        }
      } else if (hit) {
        leaf(tris, nd->data[1], nd->data[0], o, d, &r);
      }
    }
    out[k] = r;
  }
}

/* wide order: expand interior nodes (both child boxes at once), far child stacked with its tmin */
void proto_wide(const node_t *nodes, const double *tris, const double *rays, size_t n, res_t *out) {
  for (size_t k = 0; k < n; k++) {
    const double *o = rays + 6 * k, *d = o + 3;
    double inv[3] = {1.0 / d[0], 1.0 / d[1], 1.0 / d[2]};
    int sg[3] = {d[0] < 0.0, d[1] < 0.0, d[2] < 0.0};
    res_t r = {DBL_MAX, 0, 0, 0xffffffffu, 0, 0, 0, 0};
    struct { uint32_t ref; double tmin; } st[512];
    int sp = 0;
    /* the root itself: one box test (the reference's first pop) */
    double tm;
    r.nodes = 1;
    int64_t cur = -1;
    if (slab(nodes[0].bmin, nodes[0].bmax, o, inv, sg, r.t, &tm)) {
      if (nodes[0].flag) leaf(tris, nodes[0].data[1], nodes[0].data[0], o, d, &r);
      else cur = 0;
    }
    for (;;) {
      if (cur < 0) {
        if (sp == 0) break;
        --sp;
        if (!(st[sp].tmin <= r.t)) { r.pops_failed++; continue; }
        const node_t *c = nodes + st[sp].ref;
        if (c->flag) { leaf(tris, c->data[1], c->data[0], o, d, &r); continue; }
        cur = st[sp].ref;
      }
      const node_t *nd = nodes + cur;
      r.nodes += 2;
      const int o1 = sg[nd->axis], o0 = 1 - o1; /* near = data[o0], far = data[o1] */
      const node_t *cn = nodes + nd->data[o0], *cf = nodes + nd->data[o1];
      double tn, tf;
      const int hn = slab(cn->bmin, cn->bmax, o, inv, sg, r.t, &tn);
      const int hf = slab(cf->bmin, cf->bmax, o, inv, sg, r.t, &tf);
      if (hf) {
        st[sp].ref = nd->data[o1];
        st[sp].tmin = tf;
        sp++;
        if ((uint32_t)sp > r.max_sp) r.max_sp = sp;
      }
      cur = -1;
      if (hn) {
        if (cn->flag) leaf(tris, cn->data[1], cn->data[0], o, d, &r);
        else cur = nd->data[o0];
      }
    }
    out[k] = r;
  }
}
