// Host side of the device BVH builder (kernels: bvh_build.h): owns the builder's device memory, runs
// the build and the refit, and hands the engine the DWide / DLeaf arrays the wide walk reads.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>

#include "device_types.h"

struct DeviceBvh {
  // results (device memory owned by this object)
  DWide *wide = nullptr;
  DLeaf *leaves = nullptr;
  uint32_t n_wide = 0, n_leaves = 0;
  int wide_depth = 0;       // levels of the wide tree below its root
  double build_ms = 0.0;    // host wall clock of the last build(), device work included
  double refit_ms = 0.0;    // of the last refit()
  std::string error;        // message of the last call that failed

  DeviceBvh() = default;
  DeviceBvh(const DeviceBvh &) = delete;
  DeviceBvh &operator=(const DeviceBvh &) = delete;
  ~DeviceBvh();

  // Builds the tree over the n primitives at d_prims (on the current device).  d_motion_ends: 24 doubles
  // per moving primitive; d_transparency: one per primitive.  leaf_max: 1 .. 4.  Returns a hipError_t.
  int build(hipStream_t stream, const DPrim *d_prims, const double *d_motion_ends, const double *d_transparency,
            uint32_t n, uint32_t leaf_max);
  // Fits the same tree to the primitives' current boxes (after their records changed): no sort, no new
  // topology.  Returns a hipError_t.
  int refit(hipStream_t stream, const double *d_motion_ends);

 private:
  BuildArgs a_{};
  void *block_ = nullptr;  // one allocation behind every array of a_ but wide / leaves
  void release();
};
