// mgpu_kernels.hpp -- launch-side declarations shared by mgpu_kernels.hip and mgpu_api.hip
#pragma once

#include <stdint.h>

#include "../../include/mgpu.h"
#include "mgpu_device.hpp"

namespace mgpu {

constexpr int kBlock = 256;      // 4 waves per workgroup
constexpr int kChunkTiles = 2;   // 8x8-pixel tiles handed to a wave per global-counter fetch

// device-side statistics words (unsigned long long each)
enum : int { kStatTraceCalls = 0, kStatRays = 1, kStatNodes = 2, kStatTris = 3, kStatPaths = 4, kStatWords = 8 };

struct RenderParams {
  double frame[12]; // origin, corner, du, dv  (Camera::BuildCameraFrame, camera.cc:40-220)
  float plane[4];
  int has_plane;
  int W, H;         // full frame (RNG tables and hash seeds index the full frame)
  int x0, x1;       // window columns
  int y_first, strip_h, y_period, n_rows; // row strips: local row j -> y_first + (j/strip_h)*y_period + j%strip_h
  int maxPathLength, passes;
  int rng_mode;
  const uint32_t *rng_states; // device, MGPU_RNG_TABLE layout, or null
  unsigned long long seed;
  uint32_t pass_base;
  float *image;     // device, 3 * n_rows * (x1-x0)
  int32_t *count;   // device or null
  uint32_t *work_counter;        // device, zeroed before the launch
  unsigned long long *stats;     // device, kStatWords, accumulated
  double *probe;                 // device or null: kProbeStride doubles per PathTrace iteration of ONE path
  uint32_t probe_pixel, probe_pass; // full-frame pixel index and pass of the probed path
};
constexpr int kProbeStride = 16; // org[3] dir[3] t hit slot normal[3] materialID pathLength throughput.x radiance.x

// stack capacities (LDS entries per lane) the kernels are instantiated for
void launch_trace(int cap, dim3 grid, hipStream_t s, const DScene &sc, const MgpuRay *rays, size_t n,
                  MgpuIntersection *out, uint8_t *hit, unsigned long long *stats);
void launch_render(int cap, dim3 grid, hipStream_t s, const DScene &sc, const RenderParams &p);
int pick_stack_cap(int needed_entries);

} // namespace mgpu
