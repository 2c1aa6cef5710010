// sincos_turn: sin / cos of 2*pi*k/2^32 for a 32-bit draw k, as k_render_sm's cosine sampler evaluates the azimuth
// (mgpu_device.hpp): the nearest of 64 tabulated angles (the table is filled with the device library's sincospi at kernel
// start) plus a remainder |r| <= 1/64 of a half turn through two four-term series, combined by the addition theorem.
// This program compares it with sincospi(2 * k / 2^32) for EVERY k: maximal absolute difference and the maximal difference
// in units of 2^-53 (the spacing of doubles just below 1).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

#define MGPU_SINCOS_STANDALONE 1
#include "../../mallie_amd/csrc/mgpu_sincos.hpp"

__global__ void k_check(double *out) {
  __shared__ mgpu::SincosTable tbl;
  mgpu::sincos_table_fill(tbl, threadIdx.x, blockDim.x);
  __syncthreads();
  double max_abs = 0.0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < (1ull << 32); k += stride) {
    double s, c, rs, rc;
    mgpu::sincos_turn((uint32_t)k, tbl, s, c);
    sincospi(2.0 * ((double)(uint32_t)k * (1.0 / 4294967296.0)), &rs, &rc);
    const double d = fmax(fabs(s - rs), fabs(c - rc));
    max_abs = fmax(max_abs, d);
  }
  for (int off = 32; off; off >>= 1) max_abs = fmax(max_abs, __shfl_down(max_abs, off));
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)out, (unsigned long long)__double_as_longlong(max_abs));
}

int main() {
  double *d;
  hipMalloc(&d, 8);
  hipMemset(d, 0, 8);
  hipLaunchKernelGGL(k_check, dim3(2048), dim3(256), 0, 0, d);
  double h = -1;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("all 2^32 draws: max |sincos_turn - sincospi| = %.3e = %.2f units of 2^-53\n", h, h * 9007199254740992.0);
  return 0;
}
