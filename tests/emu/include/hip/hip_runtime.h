// tests/emu/include/hip/hip_runtime.h -- TEST INFRASTRUCTURE.  A stand-in for <hip/hip_runtime.h> that lets the library's .hip sources be
// compiled as plain C++ for the HOST and executed by a wave64 emulator (tests/emu/emu_runtime.cc): every workgroup is BLOCK fibers,
// 64 per wave; a fiber runs until it reaches a cross-lane operation (__ballot, __shfl*, readfirstlane, wave barrier), a workgroup
// barrier or s_sleep, where the scheduler collects the whole wave / workgroup, evaluates the operation and lets them go on.
//
// What it is for: a CPU / GPU differential -- the kernels' LOGIC (state machines, stacks, hand-out, cross-lane
// merges, host-side launch plumbing) run against the oracle where no GPU is at hand, and under host sanitizers.  What it is not:
// a product path.  libmallie_mgpu_emu.so is built by tests/emu/build_emu.py and loaded by tests only (MALLIE_MGPU_LIB); the
// product library has no CPU fallback and fails loudly without a GPU.
//
// Semantics kept: wave64, lane-minor LDS addressing, cross-lane ops among ALL live lanes of a wave (the kernels call them from
// wave-uniform control flow; a lane that shows up at a different call site than its wave aborts the run with both sites), workgroup
// barriers, atomics (one kernel runs at a time), dynamic + static LDS per workgroup, kernel arguments by value.  Not modelled: timing,
// caches, asynchronous streams (a launch runs to completion when it is enqueued; events carry host timestamps).
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <functional>

#define MGPU_EMU 1

// ---- qualifiers ---------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static __attribute__((section("emu_lds")))
#define address_space(x) // __attribute__((address_space(3))) -> __attribute__(())

// ---- vector types ---------------------------------------------------------------------------------------------------------------
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(16) int4 { int32_t x, y, z, w; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) double2 { double x, y; };
struct alignas(8) uint2 { uint32_t x, y; };
struct alignas(8) float2 { float x, y; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- the emulator's interface to device code --------------------------------------------------------------------------------------
namespace emu {
struct Idx { unsigned x, y, z; };
struct Fiber; // one lane
struct Ctx {  // what device code sees of the running lane
  Idx thread_idx, block_idx, block_dim, grid_dim;
  unsigned char *dyn_shared;
  int lane;
};
extern thread_local Ctx *g_cur;
enum Op : int { OP_BALLOT = 1, OP_SHFL, OP_FIRST, OP_WAVE_BARRIER, OP_SYNC, OP_SLEEP };
// cross-lane exchange: the calling lane contributes `v`, gets the answer; `site` = source line of the call
unsigned long long wave_ballot(int pred, int site);
unsigned long long wave_shfl(unsigned long long v, int src_lane, int site); // value of lane src_lane (own value if that lane is not live)
unsigned long long wave_first(unsigned long long v, int site);
void wave_barrier(int site);
void block_sync(int site);
void wave_sleep();
unsigned long long clock_ticks();
// name: the launch site's kernel expression.  Kernels named in MGPU_EMU_RESIDENT (default: k_trace_server) are RESIDENT kernels: all
// their workgroups are alive at once (taking turns) and the launch returns at once, the kernel running on a thread of its own until
// it leaves -- the host talks to them through mapped memory while they run.  Everything else runs to completion inside the call.
// lane_entry refers to the launch site's arguments (valid until the call returns); owning_copy() makes an entry that owns copies of
// them -- asked for only by resident kernels, which outlive the call (their arguments are plain data).
void launch(const char *name, dim3 grid, dim3 block, size_t shmem, const std::function<void()> &lane_entry,
            const std::function<std::function<void()>()> &owning_copy);
void wait_resident(); // joins resident kernels that have left or are leaving (stream / device synchronisation)
// MGPU_EMU_ISA=<hipcc -S dump>[:...] (tests/emu/isa_interp.cc): a launch whose kernel is found in one of the dumps executes the gfx950
// INSTRUCTION STREAM hipcc made of it instead of the host-compiled C++ -- `fn` = the kernel (its symbol is the device kernel's name),
// `kernarg` = its arguments laid out as the kernel-argument segment.  Returns false when the kernel is not in a dump (or the switch is off).
bool launch_isa(const void *fn, dim3 grid, dim3 block, size_t shmem, const void *kernarg, size_t kernarg_bytes);
struct ArgPack {
  unsigned char bytes[4096];
  size_t size = 0;
  template <typename T> void put(const T &v) {
    size = (size + alignof(T) - 1) / alignof(T) * alignof(T);
    if (size + sizeof(T) > sizeof(bytes)) abort();
    memcpy(bytes + size, &v, sizeof(T));
    size += sizeof(T);
  }
};
// the arguments converted to the kernel's PARAMETER types, packed with their natural alignment (= the kernel-argument segment's layout)
template <typename... P, typename... A> ArgPack pack_args(void (*)(P...), const A &...a) {
  ArgPack p;
  memset(p.bytes, 0, sizeof(p.bytes));
  (p.put<P>(static_cast<P>(a)), ...);
  return p;
}
template <typename... P, typename... A> bool try_isa(void (*k)(P...), dim3 grid, dim3 block, size_t shmem, const A &...a) {
  static const bool on = getenv("MGPU_EMU_ISA") != nullptr;
  if (!on) return false;
  const ArgPack p = pack_args(k, a...);
  return launch_isa(reinterpret_cast<const void *>(k), grid, block, shmem, p.bytes, p.size);
}
} // namespace emu

#define threadIdx (emu::g_cur->thread_idx)
#define blockIdx (emu::g_cur->block_idx)
#define blockDim (emu::g_cur->block_dim)
#define gridDim (emu::g_cur->grid_dim)

// ---- cross-lane intrinsics ------------------------------------------------------------------------------------------------------
#define __ballot(p) emu::wave_ballot((p) ? 1 : 0, __LINE__)
#define __all(p) (emu::wave_ballot((p) ? 0 : 1, __LINE__) == 0ull)
#define __any(p) (emu::wave_ballot((p) ? 1 : 0, __LINE__) != 0ull)
#define __syncthreads() emu::block_sync(__LINE__)
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier(__LINE__)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_sleep(n) emu::wave_sleep()
#define __builtin_amdgcn_s_memtime() emu::clock_ticks()
#define __threadfence() ((void)0)
#define __threadfence_block() ((void)0)

namespace emu {
template <typename T> inline unsigned long long to_bits(T v) {
  static_assert(sizeof(T) <= 8, "shuffle of a type wider than 8 bytes");
  unsigned long long b = 0;
  memcpy(&b, &v, sizeof(T));
  return b;
}
template <typename T> inline T from_bits(unsigned long long b) {
  T v;
  memcpy(&v, &b, sizeof(T));
  return v;
}
template <typename T> inline T shfl(T v, int src, int site) { return from_bits<T>(wave_shfl(to_bits(v), src, site)); }
template <typename T> inline T shfl_down(T v, int d, int site) {
  const int src = g_cur->lane + d;
  return from_bits<T>(wave_shfl(to_bits(v), src < 64 ? src : g_cur->lane, site));
}
template <typename T> inline T shfl_up(T v, int d, int site) {
  const int src = g_cur->lane - d;
  return from_bits<T>(wave_shfl(to_bits(v), src >= 0 ? src : g_cur->lane, site));
}
template <typename T> inline T shfl_xor(T v, int m, int site) { return from_bits<T>(wave_shfl(to_bits(v), g_cur->lane ^ m, site)); }
} // namespace emu
#define __shfl(v, s) emu::shfl((v), (int)(s), __LINE__)
#define __shfl_down(v, d) emu::shfl_down((v), (int)(d), __LINE__)
#define __shfl_up(v, d) emu::shfl_up((v), (int)(d), __LINE__)
#define __shfl_xor(v, m) emu::shfl_xor((v), (int)(m), __LINE__)
#define __builtin_amdgcn_readfirstlane(v) ((int)emu::wave_first((unsigned long long)(unsigned)(v), __LINE__))

// ---- bit tricks, conversions -----------------------------------------------------------------------------------------------------
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline long long __double_as_longlong(double d) { return emu::from_bits<long long>(emu::to_bits(d)); }
static inline double __longlong_as_double(long long v) { return emu::from_bits<double>((unsigned long long)v); }
static inline uint32_t __float_as_uint(float f) { return emu::from_bits<uint32_t>(emu::to_bits(f)); }
static inline float __uint_as_float(uint32_t u) { return emu::from_bits<float>(u); }
static inline int __double2loint(double d) { return (int)(uint32_t)emu::to_bits(d); }
static inline int __double2hiint(double d) { return (int)(uint32_t)(emu::to_bits(d) >> 32); }
static inline double __hiloint2double(int hi, int lo) {
  return emu::from_bits<double>(((unsigned long long)(uint32_t)hi << 32) | (unsigned long long)(uint32_t)lo);
}
static inline float __double2float_ru(double d) { // round towards +inf
  float f = (float)d;
  if ((double)f < d) f = nextafterf(f, INFINITY);
  return f;
}
static inline float __double2float_rd(double d) { // round towards -inf
  float f = (float)d;
  if ((double)f > d) f = nextafterf(f, -INFINITY);
  return f;
}
// hardware approximations: the callers refine them to the correctly rounded result (mgpu_device.hpp); the exact value is a valid seed
static inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
static inline double __builtin_amdgcn_rsq(double x) { return 1.0 / sqrt(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_sinf(float turns) { return sinf(turns * 6.283185307179586f); } // v_sin_f32 takes turns
static inline float __builtin_amdgcn_cosf(float turns) { return cosf(turns * 6.283185307179586f); }
static inline void sincospi(double x, double *s, double *c) { // device library function; <= 1 ulp from it
  *s = sin(M_PI * x);
  *c = cos(M_PI * x);
}
static inline unsigned long long clock64() { return emu::clock_ticks(); }
static inline unsigned long long wall_clock64() { return emu::clock_ticks(); }
using std::max;
using std::min;

// ---- atomics (one kernel runs at a time, lanes run one after another: plain read-modify-write) -------------------------------------------
template <typename T, typename U> static inline T atomicAdd(T *p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <typename T, typename U> static inline T atomicMax(T *p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <typename T, typename U> static inline T atomicMin(T *p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <typename T, typename U> static inline T atomicExch(T *p, U v) { T o = *p; *p = (T)v; return o; }
template <typename T, typename U, typename V> static inline T atomicCAS(T *p, U cmp, V v) { T o = *p; if (o == (T)cmp) *p = (T)v; return o; }
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
// memory the HOST may be touching meanwhile (the trace server's mailbox): real atomics
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
template <typename T> static inline T emu_atomic_fetch_max(T *p, T v) {
  T o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return o;
}
#define __hip_atomic_fetch_max(p, v, order, scope) emu_atomic_fetch_max((p), (v))

// ---- runtime API (synchronous: a launch runs when it is enqueued) -----------------------------------------------------------------
typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidConfiguration = 9, hipErrorAssert = 710,
             hipErrorUnknown = 999, hipErrorNotReady = 600 };
typedef struct emuStream *hipStream_t;
typedef struct emuEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum : unsigned { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
struct hipDeviceProp_t {
  char name[256];
  int multiProcessorCount;
  size_t totalGlobalMem;
  size_t sharedMemPerBlock;
  char gcnArchName[256];
};
hipError_t hipMalloc(void **p, size_t bytes);
template <typename T> static inline hipError_t hipMalloc(T **p, size_t bytes) { return hipMalloc((void **)p, bytes); }
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags = 0);
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned flags = 0) { return hipHostMalloc((void **)p, bytes, flags); }
hipError_t hipHostFree(void *p);
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned flags);
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s = nullptr);
hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind);
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s = nullptr);
hipError_t hipMemset(void *p, int v, size_t bytes);
hipError_t hipMemsetAsync(void *p, int v, size_t bytes, hipStream_t s = nullptr);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipDeviceSynchronize();
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipFuncSetAttribute(const void *f, hipFuncAttribute a, int v);

// the kernel's name is not allowed to contain a top-level comma (the library's launch sites bind template-ids to a variable first)
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  (emu::try_isa(kern, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__) \
       ? (void)0 \
       : emu::launch(#kern, dim3(grid), dim3(block), (size_t)(shmem), [&]() { kern(__VA_ARGS__); }, \
                     [&]() { return std::function<void()>([=]() mutable { kern(__VA_ARGS__); }); }))
