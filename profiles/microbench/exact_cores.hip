// exact_cores.hip -- are sqrt_core / rcp_core (mgpu_device.hpp: the compiler's own fp64 sqrt / division expansions without
// their range-scaling and special-case steps) bit-identical to sqrt(x) / (1.0 / x) on the ranges the kernels use them on?
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I mallie_amd/csrc -o exact_cores profiles/microbench/exact_cores.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "mgpu_device.hpp"

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

// mode 0: random mantissa, exponent uniform in [-elo, ehi]; mode 1: 1 - k * 2^-32 (the sampler's sqrt argument) and
// 1 - x^2 of its root; mode 2: near powers of two / all-ones mantissas
__global__ void k_check(uint64_t seed, int mode, int elo, int ehi, unsigned long long *bad, unsigned long long *n) {
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long b = 0, cnt = 0;
  for (int it = 0; it < 4096; ++it) {
    const uint64_t r = mix(seed + id * 4096 + it);
    double x;
    if (mode == 0) {
      const int e = elo + (int)((r >> 52) % (uint64_t)(ehi - elo + 1));
      const uint64_t bits = ((uint64_t)(e + 1023) << 52) | (r & 0xFFFFFFFFFFFFFull);
      x = __longlong_as_double((long long)bits);
    } else if (mode == 1) {
      const double u1 = (double)(uint32_t)r * (1.0 / 4294967296.0);
      const double a = 1.0 - u1; // in [2^-32, 1]
      const double ra = sqrt(a);
      if (mgpu::sqrt_core(a) != ra) ++b;
      ++cnt;
      x = fma(-ra, ra, 1.0);
      if (!(x > 0.0)) continue; // exactly 0 takes the literal form in the kernel
    } else {
      const int e = elo + (int)((r >> 52) % (uint64_t)(ehi - elo + 1));
      uint64_t m = (r & 1) ? 0xFFFFFFFFFFFFFull : 0ull;
      m ^= (r >> 1) & 0x3ull; // a few ulps around all-ones / all-zeros mantissas
      x = __longlong_as_double((long long)(((uint64_t)(e + 1023) << 52) | m));
    }
    const double s0 = sqrt(x), s1 = mgpu::sqrt_core(x);
    const double q0 = 1.0 / x, q1 = mgpu::rcp_core(x);
    const double q2 = 1.0 / -x, q3 = mgpu::rcp_core(-x);
    if (__double_as_longlong(s0) != __double_as_longlong(s1)) ++b;
    if (__double_as_longlong(q0) != __double_as_longlong(q1)) ++b;
    if (__double_as_longlong(q2) != __double_as_longlong(q3)) ++b;
    cnt += 3;
  }
  atomicAdd(bad, b);
  atomicAdd(n, cnt);
}

int main() {
  unsigned long long *d, h[2];
  hipMalloc(&d, 16);
  struct { int mode, elo, ehi; const char *what; } cases[] = {
      {0, -400, 400, "random mantissa, exponent -400..400 (the guarded range)"},
      {0, -40, 40, "random mantissa, exponent -40..40"},
      {0, -1, 1, "random mantissa, exponent -1..1"},
      {1, 0, 0, "sampler arguments: 1 - k/2^32 and 1 - x^2"},
      {2, -400, 400, "mantissas within 3 ulp of all-zeros / all-ones"},
  };
  for (auto &c : cases) {
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, 0x1234567ull + c.mode, c.mode, c.elo, c.ehi, d, d + 1);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-60s: %llu comparisons, %llu differ\n", c.what, h[1], h[0]);
  }
  return 0;
}
