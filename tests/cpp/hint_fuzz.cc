// hint_fuzz.cc -- the DEVICE's own leaf-hint code (mallie_amd/csrc/mgpu_device.hpp: leaf_hint_make, leaf_hint_apply, slab_hit_f32)
// compiled for the host through the tests' stand-in <hip/hip_runtime.h> (tests/emu/include, -DMGPU_EMU) and fuzzed against a literal
// restatement of the reference's TestLeafNode / TriangleIsect loop (bvh_accel.cc:595-697) written here, independently of the kernels.
//
// For every random leaf (4..15 triangles, in a fixed order = the reference's) and every ray that MAY consult a hint (plain, origin
// within Q of the scene's centre, |d| <= 1 + 2^-10) the in-order loop over the whole run and the in-order loop over what
// leaf_hint_apply leaves of it must end with the same (t, u, v, slot) bit for bit -- a dropped triangle is one the reference would
// not have accepted at its turn.  Rays are aimed where the rule can fail: just outside / inside triangle edges, tilted so that
// |det| sits at 0.5 .. 30 x the reference's 1024 eps guard, from far away, with and without an earlier best t.
// (The round-5 review fuzzed tests/hint_family.py's PYTHON restatement of the rule; this ties the device code itself to the claim.)
//
//   hint_fuzz [rays_in_millions = 16] [threads = hardware] [round4]  ->  one summary line; exit 1 on any difference
//   (round4: the control -- the same rays against the round-4 pads; exit 0 iff it DOES find accepted-but-dropped tests)
#include <hip/hip_runtime.h>

#include "../../mallie_amd/csrc/mgpu_device.hpp"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

namespace {

struct Tri {
  double p0[3], e1[3], e2[3];
};
struct Best {
  double t, u, v;
  uint32_t slot;
};

// TriangleIsect (bvh_accel.cc:595-638) on triangle k of the run, the reference's operations in the reference's order; updates `b`
// like TestLeafNode (bvh_accel.cc:660-690) does on an accepted test.  Returns whether it accepted.
bool reference_test(const Tri &tr, uint32_t k, const double o[3], const double d[3], Best &b) {
  const double px = d[1] * tr.e2[2] - d[2] * tr.e2[1], py = d[2] * tr.e2[0] - d[0] * tr.e2[2], pz = d[0] * tr.e2[1] - d[1] * tr.e2[0];
  const double det = tr.e1[0] * px + tr.e1[1] * py + tr.e1[2] * pz;
  if (std::fabs(det) < 2.220446049250313e-16 * 1024) return false;
  const double inv = 1.0 / det;
  const double sx = o[0] - tr.p0[0], sy = o[1] - tr.p0[1], sz = o[2] - tr.p0[2];
  const double qx = sy * tr.e1[2] - sz * tr.e1[1], qy = sz * tr.e1[0] - sx * tr.e1[2], qz = sx * tr.e1[1] - sy * tr.e1[0];
  const double u = (sx * px + sy * py + sz * pz) * inv;
  const double v = (qx * d[0] + qy * d[1] + qz * d[2]) * inv;
  const double t = (tr.e2[0] * qx + tr.e2[1] * qy + tr.e2[2] * qz) * inv;
  if (u < 0.0 || u > 1.0) return false;
  if (v < 0.0 || u + v > 1.0) return false;
  if (t < 0.0 || t > b.t) return false;
  b.t = t; b.u = u; b.v = v; b.slot = k;
  return true;
}

struct Totals {
  std::atomic<unsigned long long> rays{0}, consulted{0}, leaves{0}, leaves_hinted{0}, dropped{0}, tests{0}, accepted{0}, near_guard{0}, diffs{0}, wrongly_dropped{0},
      cone_halves{0}, bigpad_halves{0};
};

bool g_round4_rule = false; // control: the round-4 pad (2^-8 of the half's extent, no cone), which the round-4 review proved unsound

void worker(unsigned seed, unsigned long long n_rays, Totals &T) {
  std::mt19937_64 g(seed);
  auto U = [&](double a, double b) { return a + (b - a) * std::generate_canonical<double, 53>(g); };
  auto N = [&]() { return std::normal_distribution<double>(0.0, 1.0)(g); };
  unsigned long long rays = 0, consulted = 0, leaves = 0, hinted = 0, dropped = 0, tests = 0, accepted = 0, near = 0, diffs = 0, wrong = 0, cone = 0, bigpad = 0;
  while (rays < n_rays) {
    // ---- a leaf ----
    const int kind = (int)(g() % 4); // 0: random soup of mixed sizes, 1: a small mesh patch (shared edges, similar normals), 2: a wall (coplanar), 3: slivers / obtuse
    const uint32_t n = 4u + (uint32_t)(g() % 12);
    std::vector<Tri> run(n);
    const double size = std::pow(10.0, U(-1.0, 1.3));
    double base[3] = {U(-5, 5), U(-5, 5), U(-5, 5)};
    double wn[3] = {N(), N(), N()}, wa[3], wb[3]; // a wall's frame
    {
      double l = std::sqrt(wn[0] * wn[0] + wn[1] * wn[1] + wn[2] * wn[2]);
      for (double &x : wn) x /= l;
      double h[3] = {std::fabs(wn[0]) < 0.6 ? 1.0 : 0.0, std::fabs(wn[0]) < 0.6 ? 0.0 : 1.0, 0.0};
      wa[0] = wn[1] * h[2] - wn[2] * h[1]; wa[1] = wn[2] * h[0] - wn[0] * h[2]; wa[2] = wn[0] * h[1] - wn[1] * h[0];
      l = std::sqrt(wa[0] * wa[0] + wa[1] * wa[1] + wa[2] * wa[2]);
      for (double &x : wa) x /= l;
      wb[0] = wn[1] * wa[2] - wn[2] * wa[1]; wb[1] = wn[2] * wa[0] - wn[0] * wa[2]; wb[2] = wn[0] * wa[1] - wn[1] * wa[0];
    }
    for (uint32_t k = 0; k < n; ++k) {
      Tri &t = run[k];
      const double s = kind == 0 ? size * std::pow(10.0, U(-1.0, 0.5)) : size;
      if (kind == 2) { // coplanar: points a wa + b wb
        const double a0 = U(-2, 2) * s, b0 = U(-2, 2) * s, a1 = U(-1, 1) * s, b1 = U(-1, 1) * s, a2 = U(-1, 1) * s, b2 = U(-1, 1) * s;
        for (int c = 0; c < 3; ++c) {
          t.p0[c] = base[c] + a0 * wa[c] + b0 * wb[c];
          t.e1[c] = a1 * wa[c] + b1 * wb[c];
          t.e2[c] = a2 * wa[c] + b2 * wb[c];
        }
      } else if (kind == 1) { // a patch: a bumpy height field over the wall's frame
        const double a0 = U(-2, 2) * s, b0 = U(-2, 2) * s;
        for (int c = 0; c < 3; ++c) {
          t.p0[c] = base[c] + a0 * wa[c] + b0 * wb[c] + U(-0.2, 0.2) * s * wn[c];
          t.e1[c] = U(0.3, 1) * s * wa[c] + U(-0.3, 0.3) * s * wb[c] + U(-0.3, 0.3) * s * wn[c];
          t.e2[c] = U(-0.3, 0.3) * s * wa[c] + U(0.3, 1) * s * wb[c] + U(-0.3, 0.3) * s * wn[c];
        }
      } else {
        for (int c = 0; c < 3; ++c) {
          t.p0[c] = base[c] + U(-2, 2) * size;
          t.e1[c] = N() * s;
          t.e2[c] = N() * s;
        }
        if (kind == 3 && (g() & 1)) { // sliver: e2 nearly parallel to e1
          const double f = U(0.2, 1.5), eps = std::pow(10.0, U(-6, -1));
          for (int c = 0; c < 3; ++c) t.e2[c] = t.e1[c] * f + t.e2[c] * eps;
        }
      }
    }
    // ---- the scene around it: centre c, half diagonal rho_s >= the leaf's, a camera; Q as render_frames_impl derives it ----
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (const Tri &t : run)
      for (int c = 0; c < 3; ++c)
        for (double p : {t.p0[c], t.p0[c] + t.e1[c], t.p0[c] + t.e2[c]}) {
          lo[c] = std::fmin(lo[c], p);
          hi[c] = std::fmax(hi[c], p);
        }
    const double grow = std::pow(10.0, U(0.0, 1.5)); // the scene is 1 .. 30 leaves wide
    double cen[3], r2 = 0.0;
    for (int c = 0; c < 3; ++c) {
      const double w = (hi[c] - lo[c]) * grow, off = U(-0.5, 0.5) * (w - (hi[c] - lo[c]));
      const double slo = 0.5 * (lo[c] + hi[c]) + off - 0.5 * w, shi = slo + w;
      cen[c] = 0.5 * slo + 0.5 * shi;
      const double h = std::fmax(shi - cen[c], cen[c] - slo);
      r2 += h * h;
    }
    const double rho_s = std::sqrt(r2) * (1.0 + 0x1p-40);
    const double eye_dist = rho_s * std::pow(10.0, U(-0.3, 1.5));
    const double Q = std::fmax(eye_dist, rho_s) * (1.0 + 0x1p-20), Q2 = Q * Q, hint_q = Q * (1.0 + 0x1p-20);
    float rec[mgpu::kHintFloats];
    const bool has = mgpu::leaf_hint_make([&](uint32_t k) { return (const double *)&run[k]; }, n, 0.85, cen, hint_q, rec);
    ++leaves;
    if (!has) {
      rays += 16; // (a leaf without a record: nothing to check; keep the generator moving)
      continue;
    }
    ++hinted;
    if (g_round4_rule) { // the same split, boxes padded as round 4 padded them, no cone clause
      uint32_t m4;
      std::memcpy(&m4, &rec[20], 4);
      for (int part = 0; part < 2; ++part) {
        double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300}, ext = 0.0, big = 0.0;
        for (uint32_t k = part ? m4 : 0u; k < (part ? n : m4); ++k)
          for (int c = 0; c < 3; ++c)
            for (double p : {run[k].p0[c], run[k].p0[c] + run[k].e1[c], run[k].p0[c] + run[k].e2[c]}) {
              blo[c] = std::fmin(blo[c], p);
              bhi[c] = std::fmax(bhi[c], p);
            }
        for (int c = 0; c < 3; ++c) {
          ext = std::fmax(ext, bhi[c] - blo[c]);
          big = std::fmax(big, std::fmax(std::fabs(blo[c]), std::fabs(bhi[c])));
        }
        const double pad = ext * 0x1p-8 + big * 0x1p-20;
        for (int c = 0; c < 3; ++c) {
          rec[6 * part + c] = __double2float_rd(blo[c] - pad);
          rec[6 * part + 3 + c] = __double2float_ru(bhi[c] + pad);
        }
        rec[12 + 4 * part + 3] = 0.0f;
      }
    }
    cone += (rec[15] > 0.0f) + (rec[19] > 0.0f);
    bigpad += (rec[15] == 0.0f) + (rec[19] == 0.0f);
    uint32_t m;
    std::memcpy(&m, &rec[20], 4);
    float pk[mgpu::kHintFloats];
    mgpu::leaf_hint_pack(rec, pk); // the layout the kernel keeps in LDS
    const float4 f0 = make_float4(pk[0], pk[1], pk[2], pk[3]), f1 = make_float4(pk[4], pk[5], pk[6], pk[7]), f2 = make_float4(pk[8], pk[9], pk[10], pk[11]),
                 cA = make_float4(pk[12], pk[13], pk[14], pk[15]), cB = make_float4(pk[16], pk[17], pk[18], pk[19]);
    // ---- rays at this leaf ----
    for (int r = 0; r < 256; ++r, ++rays) {
      const Tri &tt = run[g() % n];
      double e1l = 0, e2l = 0, nn[3];
      nn[0] = tt.e1[1] * tt.e2[2] - tt.e1[2] * tt.e2[1]; nn[1] = tt.e1[2] * tt.e2[0] - tt.e1[0] * tt.e2[2]; nn[2] = tt.e1[0] * tt.e2[1] - tt.e1[1] * tt.e2[0];
      const double nl = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
      for (int c = 0; c < 3; ++c) { e1l += tt.e1[c] * tt.e1[c]; e2l += tt.e2[c] * tt.e2[c]; }
      e1l = std::sqrt(e1l); e2l = std::sqrt(e2l);
      if (!(nl > 0.0)) continue;
      // a target point: inside the triangle, or up to 15 % outside one of its edges
      double a = U(-0.15, 1.15), b = U(-0.15, 1.15) * (1.0 - std::fmin(std::fmax(a, 0.0), 1.0));
      double y[3], d[3], o[3];
      for (int c = 0; c < 3; ++c) y[c] = tt.p0[c] + a * tt.e1[c] + b * tt.e2[c];
      const int aim = (int)(g() % 4); // 0: any direction, 1..3: grazing the triangle's plane at the determinant guard
      if (aim == 0) {
        double l;
        do {
          for (double &x : d) x = N();
          l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        } while (!(l > 1e-3));
        for (double &x : d) x /= l;
      } else {
        // in-plane direction + a tilt psi with |det| = |e1 x e2| |d . n^| = nl sin(psi) at (0.5 .. 30) x 1024 eps
        const double w1 = N(), w2 = N();
        double ip[3], l = 0;
        for (int c = 0; c < 3; ++c) { ip[c] = w1 * tt.e1[c] / e1l + w2 * tt.e2[c] / e2l; l += ip[c] * ip[c]; }
        l = std::sqrt(l);
        if (!(l > 1e-6)) continue;
        const double psi = U(0.5, 30.0) * 2.220446049250313e-16 * 1024 / nl * ((g() & 1) ? 1.0 : -1.0);
        l = 0;
        for (int c = 0; c < 3; ++c) { d[c] = ip[c] / std::sqrt(w1 * w1 + w2 * w2 + 1e-300) + psi * nn[c] / nl; l += d[c] * d[c]; }
        l = std::sqrt(l);
        for (double &x : d) x /= l;
      }
      if (g() % 8 == 0) { // a direction as long as the kernel's may get (shading normals up to 1 + 2^-11 long)
        const double f = 1.0 + U(0.0, 0x1p-10);
        for (double &x : d) x *= f;
      }
      const double dist = std::pow(10.0, U(-1.0, 0.0)) * Q; // how far back the origin sits: up to the launch's reach
      for (int c = 0; c < 3; ++c) o[c] = y[c] - dist * d[c];
      const double ox = o[0] - cen[0], oy = o[1] - cen[1], oz = o[2] - cen[2];
      // the kernel's own permission test (mgpu_render_sm.hip, where a ray is armed)
      const double ix = 1.0 / d[0], iy = 1.0 / d[1], iz = 1.0 / d[2];
      const double lo_i = 0x1p-400, hi_i = 0x1p+400;
      const bool inv_ok = std::fabs(ix) > lo_i && std::fabs(ix) < hi_i && std::fabs(iy) > lo_i && std::fabs(iy) < hi_i && std::fabs(iz) > lo_i && std::fabs(iz) < hi_i;
      const bool may = inv_ok && std::isfinite(o[0]) && std::isfinite(o[1]) && std::isfinite(o[2]) && ox * ox + oy * oy + oz * oz <= Q2 && std::fabs(ix) < 0x1p100 &&
                       std::fabs(iy) < 0x1p100 && std::fabs(iz) < 0x1p100;
      if (!may) continue;
      // best t on arrival: none yet, or something found in an earlier leaf -- in front of, around or behind this one
      Best b0{mgpu::kDblMax, 0.0, 0.0, mgpu::kNoHit};
      if (g() % 3 == 0) b0.t = dist * U(0.5, 1.5), b0.slot = 0xFFFFu;
      // (a) the reference: every triangle, in order
      Best ref = b0;
      std::vector<char> acc(n, 0);
      for (uint32_t k = 0; k < n; ++k) {
        acc[k] = reference_test(run[k], k, o, d, ref) ? 1 : 0;
        accepted += acc[k];
      }
      {
        const double px = d[1] * tt.e2[2] - d[2] * tt.e2[1], py = d[2] * tt.e2[0] - d[0] * tt.e2[2], pz = d[0] * tt.e2[1] - d[1] * tt.e2[0];
        const double det = std::fabs(tt.e1[0] * px + tt.e1[1] * py + tt.e1[2] * pz);
        near += det >= 2.220446049250313e-16 * 1024 && det < 32.0 * 2.220446049250313e-16 * 1024;
      }
      tests += n;
      // (b) the device's consultation, then the in-order loop over what it leaves
      uint32_t cur = 0, end = n;
      const uint32_t drop = mgpu::leaf_hint_apply(f0, f1, f2, cA, cB, m, mgpu::v3(o[0], o[1], o[2]), mgpu::v3(d[0], d[1], d[2]), ix, iy, iz, b0.t, cur, end);
      ++consulted;
      dropped += drop;
      if (drop != n - (end - cur) || cur > end || end > n) { ++diffs; continue; }
      Best got = b0;
      for (uint32_t k = cur; k < end; ++k) reference_test(run[k], k, o, d, got);
      for (uint32_t k = 0; k < n; ++k)
        if ((k < cur || k >= end) && acc[k]) ++wrong;
      if (std::memcmp(&got.t, &ref.t, 8) || std::memcmp(&got.u, &ref.u, 8) || std::memcmp(&got.v, &ref.v, 8) || got.slot != ref.slot) ++diffs;
    }
  }
  T.rays += rays; T.consulted += consulted; T.leaves += leaves; T.leaves_hinted += hinted; T.dropped += dropped; T.tests += tests; T.accepted += accepted;
  T.near_guard += near; T.diffs += diffs; T.wrongly_dropped += wrong; T.cone_halves += cone; T.bigpad_halves += bigpad;
}

} // namespace

int main(int argc, char **argv) {
  const double millions = argc > 1 ? atof(argv[1]) : 16.0;
  unsigned nt = argc > 2 ? (unsigned)atoi(argv[2]) : std::thread::hardware_concurrency();
  g_round4_rule = argc > 3 && !strcmp(argv[3], "round4");
  if (nt < 1) nt = 1;
  Totals T;
  std::vector<std::thread> th;
  for (unsigned i = 0; i < nt; ++i) th.emplace_back(worker, 1000u + i, (unsigned long long)(millions * 1e6 / nt), std::ref(T));
  for (auto &t : th) t.join();
  if (g_round4_rule) printf("(control: round-4 pads, no cone) ");
  printf("hint_fuzz: %llu leaves (%llu with a record: %llu cone halves, %llu big-pad halves), %llu rays consulted, %llu tests of which %llu accepted by the reference, "
         "%llu rays within 32x of the determinant guard, %llu tests dropped by the hints (%.1f %%), accepted-but-dropped %llu, results differing %llu\n",
         (unsigned long long)T.leaves, (unsigned long long)T.leaves_hinted, (unsigned long long)T.cone_halves, (unsigned long long)T.bigpad_halves,
         (unsigned long long)T.consulted, (unsigned long long)T.tests, (unsigned long long)T.accepted, (unsigned long long)T.near_guard, (unsigned long long)T.dropped,
         100.0 * (double)T.dropped / (double)std::max<unsigned long long>(T.tests, 1), (unsigned long long)T.wrongly_dropped, (unsigned long long)T.diffs);
  if (g_round4_rule) return T.wrongly_dropped > 0 ? 0 : 1; // the control must FIND the round-4 rule's failures
  return (T.diffs || T.wrongly_dropped || T.consulted < 1000) ? 1 : 0;
}
