// mailbox_latency.hip -- what one host -> resident kernel -> host round trip costs, by where the request word lives.
//   hipcc --offload-arch=gfx950 -O3 mailbox_latency.hip -o mailbox_latency && ./mailbox_latency
// A resident wave polls a request word and answers by writing an acknowledgement word into host memory the device maps; the host
// writes the next request as soon as it sees the answer.  (a) request word in mapped host memory: the device's poll is a PCIe read;
// (b) request word in device memory the host writes through the BAR (fine-grained allocation), the device polls its own HBM.
// `payload` extra 8-byte words are read by the device from the request side before it answers (a ray is 6).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <csetjmp>
#include <csignal>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_pong(const unsigned *req, const unsigned long long *payload, int n_payload, unsigned *ack, unsigned long long *sink,
                       unsigned last, unsigned long long life_ticks) {
  const unsigned long long t0 = wall_clock64();
  unsigned seen = 0;
  unsigned long long acc = 0;
  while (seen != last && wall_clock64() - t0 < life_ticks) {
    const unsigned r = __hip_atomic_load(req, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (r != seen) {
      for (int i = 0; i < n_payload; i++) acc += __hip_atomic_load(payload + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      seen = r;
      __hip_atomic_store(ack, r + (unsigned)(acc & 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  *sink = acc;
}

static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }

static int run(const char *name, volatile unsigned *req_host, unsigned *req_dev, unsigned long long *pay_host, unsigned long long *pay_dev,
               int n_payload, int rounds) {
  unsigned *ack_host = nullptr, *ack_dev = nullptr;
  unsigned long long *sink = nullptr;
  CK(hipHostMalloc((void **)&ack_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void **)&ack_dev, ack_host, 0));
  CK(hipMalloc((void **)&sink, 8));
  *ack_host = 0;
  *req_host = 0;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipLaunchKernelGGL(k_pong, dim3(1), dim3(64), 0, st, req_dev, pay_dev, n_payload, ack_dev, sink, (unsigned)rounds, 300000000ull /* 3 s */);
  CK(hipGetLastError());
  // first round outside the clock: the kernel has to start
  const auto t00 = std::chrono::steady_clock::now();
  auto t0 = t00;
  for (int i = 1; i <= rounds; i++) {
    for (int k = 0; k < n_payload; k++) pay_host[k] = (unsigned long long)i + k;
    __atomic_store_n((unsigned *)req_host, (unsigned)i, __ATOMIC_RELEASE);
    while (__atomic_load_n(ack_host, __ATOMIC_ACQUIRE) != (unsigned)i) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t00).count() > 5.0) { printf("%s: no answer at round %d\n", name, i); return 1; }
    }
    if (i == 1) t0 = std::chrono::steady_clock::now();
  }
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (rounds - 1);
  CK(hipStreamSynchronize(st));
  printf("%-58s payload %d words: %.2f us per round trip\n", name, n_payload, us);
  CK(hipStreamDestroy(st));
  CK(hipHostFree(ack_host));
  CK(hipFree(sink));
  return 0;
}

int main() {
  const int rounds = 20000;
  for (int n_payload : {0, 6}) {
    unsigned *h = nullptr, *d = nullptr;
    CK(hipHostMalloc((void **)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&d, h, 0));
    if (run("(a) request in mapped host memory (device polls over PCIe)", h, d, (unsigned long long *)(h + 16), (unsigned long long *)(d + 16), n_payload, rounds)) return 1;
    CK(hipHostFree(h));
  }
  // (b) device memory written by the host: try the fine-grained allocation, then plain hipMalloc
  for (int kind = 0; kind < 2; kind++) {
    unsigned *v = nullptr;
    hipError_t e = kind == 0 ? hipExtMallocWithFlags((void **)&v, 4096, hipDeviceMallocFinegrained) : hipMalloc((void **)&v, 4096);
    if (e != hipSuccess) { printf("(b) kind %d: allocation failed: %s\n", kind, hipGetErrorString(e)); continue; }
    CK(hipMemset(v, 0, 4096));
    CK(hipDeviceSynchronize());
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    if (sigsetjmp(g_jmp, 1)) {
      printf("(b) %s: the host cannot write it (fault)\n", kind == 0 ? "hipDeviceMallocFinegrained" : "hipMalloc");
      continue;
    }
    volatile unsigned probe = v[0]; // faults here if the BAR mapping is not there
    (void)probe;
    for (int n_payload : {0, 6}) {
      const char *name = kind == 0 ? "(b) request in fine-grained device memory (host writes BAR)" : "(b) request in hipMalloc device memory (host writes BAR)";
      if (run(name, v, v, (unsigned long long *)(v + 16), (unsigned long long *)(v + 16), n_payload, rounds)) break;
    }
    signal(SIGSEGV, SIG_DFL);
    signal(SIGBUS, SIG_DFL);
    (void)hipFree(v);
  }
  return 0;
}
