// tcp_rate.hip -- how many divergent 16-byte L1 hits per cycle does one CU serve?
// Every lane loads 16 bytes; `group` consecutive lanes share one 128-byte line (group = 1: every lane its own line,
// group = 8: perfectly coalesced lines).  Working set per workgroup: `lines` lines of 128 B (L1-resident when small).
// 16 waves per CU (4 workgroups of 256), 256 CUs.  Prints lane-loads and line-accesses per CU cycle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256, 4) void k(const uint4 *base, int lines, int group, int iters, unsigned long long *out, uint4 *sink) {
  const int lane = threadIdx.x & 63;
  unsigned s = (blockIdx.x * 256 + threadIdx.x) / group * 2654435761u + 12345u;
  const uint4 *mine = base + (size_t)blockIdx.x * lines * 8;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s = s * 1664525u + 1013904223u;
      const unsigned line = (s >> 8) % (unsigned)lines;
      const uint4 v = mine[(size_t)line * 8 + (lane % group) % 8];
      acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
  }
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc.x == 0x12345678u) sink[0] = acc;
}
int main() {
  const int blocks = 256 * 4, iters = 2000;
  for (int lines : {64, 1024, 16384}) {   // 8 KB (L1), 128 KB per WG (L2), 2 MB per WG (L2/MALL)
    uint4 *d; unsigned long long *o; uint4 *sink;
    hipMalloc(&d, (size_t)blocks * lines * 128); hipMemset(d, 1, (size_t)blocks * lines * 128);
    hipMalloc(&o, blocks * 8); hipMalloc(&sink, 16);
    for (int group : {1, 2, 4, 8}) {
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, lines, group, iters, o, sink);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, lines, group, iters, o, sink);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(blocks);
      hipMemcpy(h.data(), o, blocks * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (auto v : h) avg += v; avg /= blocks;
      const double lane_loads_per_cu = 4.0 * 256 * iters * 4;   // 4 WGs per CU
      printf("lines/WG %6d group %d: %.0f cycles  lane-loads/CU-cycle %.3f  line-accesses/CU-cycle %.3f\n", lines, group, avg,
             lane_loads_per_cu / avg, lane_loads_per_cu / group / avg);
    }
    hipFree(d); hipFree(o); hipFree(sink);
  }
}
