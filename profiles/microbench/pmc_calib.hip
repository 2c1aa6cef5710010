// pmc_calib.hip -- known byte counts for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the access patterns of
// this library (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern before trusting an
// absolute").  Three kernels over a buffer far larger than the 256 MiB Infinity Cache (1 GiB), each run 3 times:
//   k_calib_read16   every lane reads 16 bytes, coalesced (what k_accumulate_tiled and the scene staging do)   -> 1 GiB read
//   k_calib_write16  every lane writes 16 bytes, coalesced (k_accumulate_tiled's image stores)                -> 1 GiB written
//   k_calib_write12  every lane writes three dwords at 12-byte pitch, all lanes of a wave at once: the stores of
//                    k_render_sm's radiance planes (one pixel = 12 bytes, tile-major) when a whole tile lands together
//                                                                                                        -> 0.75 GiB written
// Build: hipcc --offload-arch=gfx950 -O3 profiles/microbench/pmc_calib.hip -o profiles/microbench/pmc_calib
// Run:   rocprofv3 --pmc FETCH_SIZE -- profiles/microbench/pmc_calib      (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_calib_read16(const uint4 *__restrict__ src, size_t n, unsigned *sink) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint4 v = src[i];
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = 1u; // never true for the fill pattern: keeps the load alive
}

__global__ __launch_bounds__(256) void k_calib_write16(uint4 *__restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  dst[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}

__global__ __launch_bounds__(256) void k_calib_write12(float *__restrict__ dst, size_t npix) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  float *p = dst + 3 * i;
  p[0] = 1.0f;
  p[1] = 2.0f;
  p[2] = 3.0f;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  void *buf = nullptr;
  unsigned *sink = nullptr;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc((void **)&sink, 4));
  CHECK(hipMemset(buf, 0x5a, bytes));
  CHECK(hipDeviceSynchronize());
  const size_t n16 = bytes / 16, npix = bytes / 16; // write12: 12 bytes per pixel -> 0.75 GiB
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_calib_read16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const uint4 *)buf, n16, sink);
    hipLaunchKernelGGL(k_calib_write16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (uint4 *)buf, n16);
    hipLaunchKernelGGL(k_calib_write12, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, 0, (float *)buf, npix);
  }
  CHECK(hipDeviceSynchronize());
  printf("pmc_calib: read16 %zu bytes, write16 %zu bytes, write12 %zu bytes per launch, 3 launches each\n", bytes, bytes, npix * 12);
  return 0;
}
