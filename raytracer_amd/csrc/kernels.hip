// The only translation unit compiled for the GPU (gfx950): instantiates the pipeline kernels of
// pipeline.h and wraps their launches in plain C++ functions for the host-side engine.
#include <hip/hip_runtime.h>

#include "bvh_build.h"
#include "pipeline.h"

template <bool PRIMARY, bool UV, bool MOTION, int WALK>
static void launch_trace_t(unsigned grid, hipStream_t st, const TraceArgs &a) {
  hipLaunchKernelGGL((k_trace<PRIMARY, UV, MOTION, WALK>), dim3(grid), dim3(256), 0, st, a);
}

template <int WALK>
static void launch_trace_w(bool primary, bool uv, bool motion, unsigned grid, hipStream_t st, const TraceArgs &a) {
  const int v = (primary ? 4 : 0) | (uv ? 2 : 0) | (motion ? 1 : 0);
  switch (v) {
    case 0: launch_trace_t<false, false, false, WALK>(grid, st, a); break;
    case 1: launch_trace_t<false, false, true, WALK>(grid, st, a); break;
    case 2: launch_trace_t<false, true, false, WALK>(grid, st, a); break;
    case 3: launch_trace_t<false, true, true, WALK>(grid, st, a); break;
    case 4: launch_trace_t<true, false, false, WALK>(grid, st, a); break;
    case 5: launch_trace_t<true, false, true, WALK>(grid, st, a); break;
    case 6: launch_trace_t<true, true, false, WALK>(grid, st, a); break;
    default: launch_trace_t<true, true, true, WALK>(grid, st, a); break;
  }
}

// Dispatch on the walk.  MIRT_WALKS (development aid) is a bit mask of the walks to compile, bit = WALK_*;
// a walk left out falls back to the per-lane walk in the reference's order.
#ifndef MIRT_WALKS
#define MIRT_WALKS 15
#endif
#if MIRT_WALKS & 1
#define IF_PACKET(...) __VA_ARGS__
#else
#define IF_PACKET(...) (void)0
#endif
#if MIRT_WALKS & 4
#define IF_ORDERED(...) __VA_ARGS__
#else
#define IF_ORDERED(...) (void)0
#endif
#if MIRT_WALKS & 8
#define IF_WIDE(...) __VA_ARGS__
#else
#define IF_WIDE(...) (void)0
#endif
#define BY_WALK(walk, call_packet, call_lane, call_ordered, call_wide)             \
  do {                                                                             \
    if ((walk) == WALK_PACKET && (MIRT_WALKS & 1)) IF_PACKET(call_packet);         \
    else if ((walk) == WALK_ORDERED && (MIRT_WALKS & 4)) IF_ORDERED(call_ordered); \
    else if ((walk) == WALK_WIDE && (MIRT_WALKS & 8)) IF_WIDE(call_wide);          \
    else call_lane;                                                                \
  } while (0)

void launch_trace(int walk, bool primary, bool uv, bool motion, unsigned grid, void *stream, const TraceArgs &a) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  BY_WALK(walk, launch_trace_w<WALK_PACKET>(primary, uv, motion, grid, st, a),
          launch_trace_w<WALK_LANE>(primary, uv, motion, grid, st, a),
          launch_trace_w<WALK_ORDERED>(primary, uv, motion, grid, st, a),
          launch_trace_w<WALK_WIDE>(primary, uv, motion, grid, st, a));
}

template <bool STRATA> static void launch_shadow_t(int walk, bool any_hit, bool point, unsigned grid, hipStream_t st, const ShadowArgs &a) {
  if (any_hit && point && !STRATA && walk == WALK_PACKET && (MIRT_WALKS & 1)) {
    IF_PACKET(hipLaunchKernelGGL((k_shadow<WALK_PACKET, false, true, true>), dim3(grid), dim3(256), 0, st, a));
    return;
  }
  if (any_hit && walk == WALK_PACKET && (MIRT_WALKS & 1)) {
    IF_PACKET(hipLaunchKernelGGL((k_shadow<WALK_PACKET, STRATA, true>), dim3(grid), dim3(256), 0, st, a));
    return;
  }
  BY_WALK(walk, hipLaunchKernelGGL((k_shadow<WALK_PACKET, STRATA>), dim3(grid), dim3(256), 0, st, a),
          hipLaunchKernelGGL((k_shadow<WALK_LANE, STRATA>), dim3(grid), dim3(256), 0, st, a),
          hipLaunchKernelGGL((k_shadow<WALK_ORDERED, STRATA>), dim3(grid), dim3(256), 0, st, a),
          hipLaunchKernelGGL((k_shadow<WALK_WIDE, STRATA>), dim3(grid), dim3(256), 0, st, a));
}
// any_hit: the scene allows it (DScene::shadow_any_hit); used by the packet walk, ignored by the others
// point: no light is an area light (with any_hit and without strata the packet walk then runs its shortest form)
void launch_shadow(int walk, bool strata, bool any_hit, bool point, unsigned grid, void *stream, const ShadowArgs &a) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (strata) launch_shadow_t<true>(walk, any_hit, point, grid, st, a);
  else launch_shadow_t<false>(walk, any_hit, point, grid, st, a);
}

template <bool PRIMARY, bool UV, bool MOTION>
static void launch_closest_stream_t(unsigned grid, hipStream_t st, const TraceArgs &a) {
  hipLaunchKernelGGL((k_closest_stream<PRIMARY, UV, MOTION>), dim3(grid), dim3(256), 0, st, a);
}
void launch_closest_stream(bool primary, bool uv, bool motion, unsigned grid, void *stream, const TraceArgs &a) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int v = (primary ? 4 : 0) | (uv ? 2 : 0) | (motion ? 1 : 0);
  switch (v) {
    case 0: launch_closest_stream_t<false, false, false>(grid, st, a); break;
    case 1: launch_closest_stream_t<false, false, true>(grid, st, a); break;
    case 2: launch_closest_stream_t<false, true, false>(grid, st, a); break;
    case 3: launch_closest_stream_t<false, true, true>(grid, st, a); break;
    case 4: launch_closest_stream_t<true, false, false>(grid, st, a); break;
    case 5: launch_closest_stream_t<true, false, true>(grid, st, a); break;
    case 6: launch_closest_stream_t<true, true, false>(grid, st, a); break;
    default: launch_closest_stream_t<true, true, true>(grid, st, a); break;
  }
}
void launch_shade(bool primary, bool uv, unsigned grid, void *stream, const TraceArgs &a) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (primary && uv) hipLaunchKernelGGL((k_shade<true, true>), dim3(grid), dim3(256), 0, st, a);
  else if (primary) hipLaunchKernelGGL((k_shade<true, false>), dim3(grid), dim3(256), 0, st, a);
  else if (uv) hipLaunchKernelGGL((k_shade<false, true>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_shade<false, false>), dim3(grid), dim3(256), 0, st, a);
}

void launch_shadow_stream(bool any_hit, unsigned grid, void *stream, const ShadowArgs &a) {
  if (any_hit) hipLaunchKernelGGL(k_shadow_stream<true>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(k_shadow_stream<false>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
}

void launch_deinterleave(unsigned grid, void *stream, const DeinterleaveArgs &a) {
  hipLaunchKernelGGL(k_deinterleave, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
}

void launch_resolve(unsigned grid, void *stream, const ResolveArgs &a) {
  hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
}

void launch_trace_ids(int walk, unsigned grid, void *stream, const IdsArgs &a) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  BY_WALK(walk, hipLaunchKernelGGL(k_trace_ids<WALK_PACKET>, dim3(grid), dim3(256), 0, st, a),
          hipLaunchKernelGGL(k_trace_ids<WALK_LANE>, dim3(grid), dim3(256), 0, st, a),
          hipLaunchKernelGGL(k_trace_ids<WALK_ORDERED>, dim3(grid), dim3(256), 0, st, a),
          hipLaunchKernelGGL(k_trace_ids<WALK_WIDE>, dim3(grid), dim3(256), 0, st, a));
}

void launch_trace_rays(int walk, unsigned grid, void *stream, const RaysArgs &a) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  BY_WALK(walk, hipLaunchKernelGGL(k_trace_rays<WALK_PACKET>, dim3(grid), dim3(256), 0, st, a),
          hipLaunchKernelGGL(k_trace_rays<WALK_LANE>, dim3(grid), dim3(256), 0, st, a),
          hipLaunchKernelGGL(k_trace_rays<WALK_ORDERED>, dim3(grid), dim3(256), 0, st, a),
          hipLaunchKernelGGL(k_trace_rays<WALK_WIDE>, dim3(grid), dim3(256), 0, st, a));
}

// ---- working memory for the entry point that allocates nothing ----------------------------------
// One array per GPU, held from the moment the code object is loaded until the process ends.  In host
// code the array's name is only a handle for the runtime: the memory's device address is asked for
// once per device (with that device current) and everything -- kernel arguments, fills, copies --
// uses that address.  Every call on a GPU shares that GPU's array; calls that overlap on one GPU race.
static __device__ __align__(256) unsigned char scratch[MIRT_SCRATCH_BYTES];

#define MIRT_SCRATCH_MAX_DEVICES 64
static void *g_scratch_ptr[MIRT_SCRATCH_MAX_DEVICES];

// Device address of `device`'s copy of the array; `device` must be the current device.
int scratch_address(int device, void **out) {
  if (device < 0 || device >= MIRT_SCRATCH_MAX_DEVICES) return (int)hipErrorInvalidDevice;
  if (!g_scratch_ptr[device]) {
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return (int)e;
    if (!p) return (int)hipErrorInvalidSymbol;
    g_scratch_ptr[device] = p;
  }
  *out = g_scratch_ptr[device];
  return 0;
}
