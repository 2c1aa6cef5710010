#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, int iters, int mode) {
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001, c = 0.5;
  float fa = a, fb = b, fc = c;
  for (int i = 0; i < iters; i++) {
    if (mode == 0) { a = a * b + c; b = b * 0.9999999 + 1e-9; c = c * a - b; a = a - c * 0.5; }
    else { fa = fa * fb + fc; fb = fb * 0.9999999f + 1e-9f; fc = fc * fa - fb; fa = fa - fc * 0.5f; }
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  if (a + fa == 12345.678) out[2] = 1;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64); unsigned long long h[3];
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate %d kHz\n", rate);
  for (int mode = 0; mode < 2; mode++) for (int blocks : {256, 1024, 2048}) {
    k<<<blocks, 256>>>(d, 2000000, mode); hipDeviceSynchronize();
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    double secs = (double)h[1] / (rate * 1e3);
    printf("%s blocks=%d: %llu shader ticks in %.3f ms -> %.3f GHz\n", mode ? "fp32" : "fp64", blocks, h[0], secs * 1e3, h[0] / secs / 1e9);
  }
}
