// The only translation unit compiled for the GPU (gfx950): instantiates the pipeline kernels of
// pipeline.h and wraps their launches in plain C++ functions for the host-side engine.
#include <hip/hip_runtime.h>

#include "pipeline.h"

template <bool PRIMARY, bool UV, bool MOTION, bool PACKET>
static void launch_trace_t(unsigned grid, hipStream_t st, const TraceArgs &a) {
  hipLaunchKernelGGL((k_trace<PRIMARY, UV, MOTION, PACKET>), dim3(grid), dim3(256), 0, st, a);
}

template <bool PACKET>
static void launch_trace_p(bool primary, bool uv, bool motion, unsigned grid, hipStream_t st, const TraceArgs &a) {
  const int v = (primary ? 4 : 0) | (uv ? 2 : 0) | (motion ? 1 : 0);
  switch (v) {
    case 0: launch_trace_t<false, false, false, PACKET>(grid, st, a); break;
    case 1: launch_trace_t<false, false, true, PACKET>(grid, st, a); break;
    case 2: launch_trace_t<false, true, false, PACKET>(grid, st, a); break;
    case 3: launch_trace_t<false, true, true, PACKET>(grid, st, a); break;
    case 4: launch_trace_t<true, false, false, PACKET>(grid, st, a); break;
    case 5: launch_trace_t<true, false, true, PACKET>(grid, st, a); break;
    case 6: launch_trace_t<true, true, false, PACKET>(grid, st, a); break;
    default: launch_trace_t<true, true, true, PACKET>(grid, st, a); break;
  }
}

void launch_trace(bool packet, bool primary, bool uv, bool motion, unsigned grid, void *stream, const TraceArgs &a) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (packet) launch_trace_p<true>(primary, uv, motion, grid, st, a);
  else launch_trace_p<false>(primary, uv, motion, grid, st, a);
}

void launch_shadow(bool packet, unsigned grid, void *stream, const ShadowArgs &a) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (packet) hipLaunchKernelGGL(k_shadow<true>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_shadow<false>, dim3(grid), dim3(256), 0, st, a);
}

void launch_resolve(unsigned grid, void *stream, const ResolveArgs &a) {
  hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
}

void launch_trace_ids(bool packet, unsigned grid, void *stream, const IdsArgs &a) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (packet) hipLaunchKernelGGL(k_trace_ids<true>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_trace_ids<false>, dim3(grid), dim3(256), 0, st, a);
}

void launch_trace_rays(bool packet, unsigned grid, void *stream, const RaysArgs &a) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (packet) hipLaunchKernelGGL(k_trace_rays<true>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_trace_rays<false>, dim3(grid), dim3(256), 0, st, a);
}
