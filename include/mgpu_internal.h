/* mgpu_internal.h -- instrumentation entry points of libmallie_mgpu.so that are NOT part of the drop-in boundary (include/mgpu.h).
 *
 * Nothing a caller of Mallie's Scene / BVHAccel / Render API needs is declared here: these read raw counter words, dump the state
 * of the last launch or run a measurement loop inside the library, for this repository's own tests, tools/ and bench.py.  They are
 * exported by the same shared object (the tests reach them by ctypes), may change between rounds without an ABI-version bump,
 * and a binding for the reference never includes this file. */
#ifndef MGPU_INTERNAL_H
#define MGPU_INTERNAL_H

#include "mgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement aid (bench.py): rays[0..n) traced as n ONE-ray mgpu_trace calls issued by `threads` host threads of this
 * library's own (thread t takes rays t, t + threads, ...), i.e. the reference's calling pattern of Scene::Trace without a
 * binding's per-call overhead in the clock; out / hit receive the records, *calls_per_s the rate (first call outside the clock). */
int mgpu_trace_calls_measure(MgpuScene *scene, const MgpuRay *rays, size_t n, int threads, MgpuIntersection *out, uint8_t *hit,
                             double *calls_per_s);

/* Diagnostic: the cached classification of the reference-stream mode, 0 / 1 / 2 per pixel (mgpu_render_stream). */
int mgpu_debug_stream_classes(MgpuScene *scene, unsigned char *out, size_t npix); /* diagnostic: the cached classes, 0 / 1 / 2 per pixel */

/* Diagnostic: the 32 raw device counter words (layout in mallie_amd/csrc/mgpu_kernels.hpp; words 8.. are only filled by
 * -DMGPU_UTIL experiment builds). */
int mgpu_debug_words(MgpuScene *scene, unsigned long long *out32);

/* Active-lane accounting of the render kernel's three bodies (NODE: box tests, TRI: triangle tests, SHADE: the rest of a
 * PathTrace iteration), accumulated like the work counters (mgpu_stats_read resets them too).  The kernel books about one
 * step in `sample_every` (chosen by the low bits of the shader clock, whatever the step does): `*_trips` = trips of the
 * body's loop the wave made on the booked steps, `*_lanes` = lanes active summed over those trips, so
 * lanes / (64 * trips) is the body's active-lane fraction; `node_steps` / `tri_steps` / `shade_steps` = steps booked
 * (SHADE has one trip per step).  Synchronises the device. */
typedef struct {
  uint64_t node_trips, node_lanes, tri_trips, tri_lanes, shade_steps, shade_lanes, node_steps, tri_steps;
  uint32_t sample_every, pad_;
} MgpuOccupancy;
int mgpu_occupancy_read(MgpuScene *scene, MgpuOccupancy *out);

/* Diagnostic (MGPU_WAVE_LOG=1 + -DMGPU_UTIL builds): 8 words per wave, `out` holds 8 * n_waves words:
 * {start, end, time the work cursor was found dry (100 MHz device ticks), XCC id | rays << 8,
 *  rays traced after dry, lanes alive at dry | their pathLength sum << 16, NODE | TRI << 20 | SHADE << 40 steps after dry,
 *  scheduling rounds after dry}. */
int mgpu_debug_wave_log(MgpuScene *scene, unsigned long long *out, size_t n_waves);

/* Diagnostic: the cost-ordered hand-out state of the last render launch: per 8x8 tile (row-major over the rendered
 * window) the cost measured by that launch, and the order in which it handed the tiles out.  Either pointer may be NULL;
 * n_tiles must not exceed the launch's tile count.  Synchronises the device. */
int mgpu_debug_tile_order(MgpuScene *scene, uint32_t *cost_out, uint32_t *order_out, size_t n_tiles);

#ifdef __cplusplus
}
#endif

#endif /* MGPU_INTERNAL_H */
