// The only translation unit compiled for the GPU (gfx950): instantiates the pipeline kernels of
// pipeline.h and wraps their launches in plain C++ functions for the host-side engine.
#include <hip/hip_runtime.h>

#include "pipeline.h"

template <bool PRIMARY, bool UV, bool MOTION>
static void launch_trace_t(unsigned grid, hipStream_t st, const TraceArgs &a) {
  hipLaunchKernelGGL((k_trace<PRIMARY, UV, MOTION>), dim3(grid), dim3(256), 0, st, a);
}

void launch_trace(bool primary, bool uv, bool motion, unsigned grid, void *stream, const TraceArgs &a) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int v = (primary ? 4 : 0) | (uv ? 2 : 0) | (motion ? 1 : 0);
  switch (v) {
    case 0: launch_trace_t<false, false, false>(grid, st, a); break;
    case 1: launch_trace_t<false, false, true>(grid, st, a); break;
    case 2: launch_trace_t<false, true, false>(grid, st, a); break;
    case 3: launch_trace_t<false, true, true>(grid, st, a); break;
    case 4: launch_trace_t<true, false, false>(grid, st, a); break;
    case 5: launch_trace_t<true, false, true>(grid, st, a); break;
    case 6: launch_trace_t<true, true, false>(grid, st, a); break;
    default: launch_trace_t<true, true, true>(grid, st, a); break;
  }
}

void launch_shadow(unsigned grid, void *stream, const ShadowArgs &a) {
  hipLaunchKernelGGL(k_shadow, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
}

void launch_resolve(unsigned grid, void *stream, const ResolveArgs &a) {
  hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
}

void launch_trace_ids(unsigned grid, void *stream, const IdsArgs &a) {
  hipLaunchKernelGGL(k_trace_ids, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
}

void launch_trace_rays(unsigned grid, void *stream, const RaysArgs &a) {
  hipLaunchKernelGGL(k_trace_rays, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
}
