#include <hip/hip_runtime.h>
#include <cstdio>
// Does a wave64 VALU instruction cost less when only part of the wave is active?  (Round 4: decides whether compacting the
// active lanes of a divergent step into one half / one quarter of the wave could pay.)
// 8 waves per SIMD, every wave runs the same loop of 8 independent instructions with EXEC = `mask`.
#define BODY8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(fm), "v"(m))
#define I_FMA64(k) "v_fma_f64 %8, %8, %13, %13\n v_fma_f64 %9, %9, %13, %13\n v_fma_f64 %10, %10, %13, %13\n v_fma_f64 %11, %11, %13, %13\n"
#define I_FMA32(k) "v_fmac_f32 %" #k ", %" #k ", %12\n"
#define I_MAX64(k) "v_max_f64 %8, %8, %13\n v_max_f64 %9, %9, %13\n v_max_f64 %10, %10, %13\n v_max_f64 %11, %11, %13\n"
template <int MODE> __global__ void k(unsigned long long *out, int iters, float seed, unsigned long long mask) {
  float f0 = seed + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
  double a0 = f0, a1 = f1, a2 = f2, a3 = f3;
  const float fm = 1.0000001f;
  const double m = 1.0000001;
  if ((mask >> (threadIdx.x & 63)) & 1ull) {
    for (int i = 0; i < iters; i++) {
      if (MODE == 0) BODY8(I_FMA64);
      if (MODE == 1) BODY8(I_FMA32);
      if (MODE == 2) BODY8(I_MAX64);
    }
  }
  if (f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + a0 + a1 + a2 + a3 == 1.2345) out[1] = 1;
}
template <int MODE> void run(const char *name, unsigned long long *d, int per, unsigned long long mask, const char *mname) {
  int iters = 20000, wps = 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<256 * wps, 256>>>(d, 1000, 1.0f, mask);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<256 * wps, 256>>>(d, iters, 1.0f, mask);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-12s exec %-22s %.2f cycles per wave-instruction per SIMD\n", name, mname, ms * 1e-3 * 2.39e9 / ((double)per * iters * wps));
}
int main() {
  unsigned long long *d;
  hipMalloc(&d, 64);
  struct { unsigned long long m; const char *n; } masks[] = {
      {~0ull, "all 64"}, {0xffffffffull, "low 32"}, {0xffffffff00000000ull, "high 32"}, {0x5555555555555555ull, "even lanes (32)"},
      {0xffffull, "low 16"}, {0x0000ffff0000ffffull, "16 + 16 (0-15,32-47)"}, {0xffff0000ull, "lanes 16-31"}, {1ull, "lane 0"}};
  for (auto &mk : masks) {
    run<0>("v_fma_f64", d, 32, mk.m, mk.n);
    run<1>("v_fmac_f32", d, 8, mk.m, mk.n);
    run<2>("v_max_f64", d, 32, mk.m, mk.n);
  }
}
