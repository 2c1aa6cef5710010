// mgpu_sincos.hpp -- sin / cos of the azimuth 2*pi*u of the cosine sampler (SampleDiffuseIS, render.cc:325-333) straight
// from the 32-bit draw k that u = k / 2^32 was made of.
//
// The reference forms phi = 2*pi*u in double (one rounding, <= 4.4e-16 absolute) and calls glibc's sin / cos (< 1 ulp); the
// device's default has been sincospi(2u) of the device library (<= 1 ulp of the exact value, no rounding of the argument):
// two evaluations of the same two numbers that differ by up to ~1e-15, which no test has ever seen move a hit / miss decision
// (DESIGN.md 5).  This is a third such evaluation, made for the instruction count: ~20 vector instructions instead of ~75.
//   x = 2u = k / 2^31 half turns;  j = round(32 x) selects one of 64 tabulated angles j / 32 (sincospi of the device
//   library, computed when the kernel starts: <= 1 ulp), r = x - j / 32 is exact, |r| <= 1/64, and
//   sin(pi r), cos(pi r) come from four-term series in (pi r)^2 (truncation < 1e-16 relative); the addition theorem combines
//   them.  profiles/microbench/sincos_turn.hip compares it with sincospi for every one of the 2^32 draws.
// Table and coefficients live in LDS: they reach the lanes through the LDS pipe (a broadcast read), not as VALU moves of
// 64-bit literals, which is where a third of the library routine's instructions go.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mgpu {

struct alignas(16) SincosTable {
  double2 sc[64];   // (sin, cos) of pi * j / 32
  double2 coef[4];  // (S_i, C_i): sin(pi r) = r * (S0 + t (S1 + t (S2 + t S3))), cos(pi r) = 1 + t (C0 + t (C1 + t (C2 + t C3))), t = r^2
};

// called by the first 64+ threads of a workgroup before a barrier
__device__ __forceinline__ void sincos_table_fill(SincosTable &t, unsigned tid, unsigned nthreads) {
  for (unsigned j = tid; j < 64; j += nthreads) {
    double s, c;
    sincospi((double)j * (1.0 / 32.0), &s, &c);
    t.sc[j] = make_double2(s, c);
  }
  if (tid == 0) { // (-1)^i pi^(2i+1) / (2i+1)!  and  (-1)^(i+1) pi^(2i+2) / (2i+2)!
    t.coef[0] = make_double2(3.141592653589793238, -4.934802200544679310);
    t.coef[1] = make_double2(-5.167712780049970029, 4.058712126416768218);
    t.coef[2] = make_double2(2.550164039877345443, -1.335262768854589495);
    t.coef[3] = make_double2(-0.599264529320792077, 0.235330630358893205);
  }
}

__device__ __forceinline__ void sincos_turn(uint32_t k, const SincosTable &tb, double &s, double &c) {
  const uint32_t j = (k + (1u << 25)) >> 26;                 // nearest tabulated angle, 0 .. 64 (64 = a full turn = 0)
  const int32_t ri = (int32_t)(k - (j << 26));               // remainder in units of 2^-31 half turns, |ri| <= 2^25
  const double r = (double)ri * (1.0 / 2147483648.0);        // exact
  const double2 a = tb.sc[j & 63u];
  const double2 k0 = tb.coef[0], k1 = tb.coef[1], k2 = tb.coef[2], k3 = tb.coef[3];
  const double t = r * r;
  const double sr = r * fma(fma(fma(k3.x, t, k2.x), t, k1.x), t, k0.x);
  const double cr = fma(fma(fma(fma(k3.y, t, k2.y), t, k1.y), t, k0.y), t, 1.0);
  s = fma(a.x, cr, a.y * sr);
  c = fma(a.y, cr, -(a.x * sr));
}

} // namespace mgpu
