// Host side of the device BVH builder: see bvh_device.h, and bvh_build.h for the kernels.
#include "bvh_device.h"

#include <chrono>
#include <cmath>
#include <cstring>

namespace {

#define BVH_TRY(expr)                                             \
  do {                                                            \
    hipError_t e_ = (expr);                                       \
    if (e_ != hipSuccess) {                                       \
      error = std::string(#expr) + ": " + hipGetErrorString(e_);  \
      return static_cast<int>(e_);                                \
    }                                                             \
  } while (0)

size_t padded(size_t bytes) { return (bytes + 255) & ~static_cast<size_t>(255); }

double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

DeviceBvh::~DeviceBvh() { release(); }

void DeviceBvh::release() {
  if (block_) (void)hipFree(block_);
  if (wide) (void)hipFree(wide);
  if (leaves) (void)hipFree(leaves);
  block_ = nullptr;
  wide = nullptr;
  leaves = nullptr;
  n_wide = n_leaves = 0;
}

int DeviceBvh::build(hipStream_t stream, const DPrim *d_prims, const double *d_motion_ends,
                     const double *d_transparency, uint32_t n, uint32_t leaf_max) {
  const auto t0 = std::chrono::steady_clock::now();
  (void)hipGetLastError();  // what an earlier, unrelated call of this thread left behind is not this build's error
  release();
  std::memset(&a_, 0, sizeof a_);
  a_.prims = d_prims;
  a_.motion_ends = d_motion_ends;
  a_.transparency = d_transparency;
  a_.n = n;
  a_.leaf_max = leaf_max < 1 ? 1 : (leaf_max > 4 ? 4 : leaf_max);
  a_.sort_blocks = n ? (n + BUILD_SORT_TILE - 1) / BUILD_SORT_TILE : 1;

  const size_t n1 = n ? n : 1, inner = n > 1 ? n - 1 : 1;
  BVH_TRY(hipMalloc(reinterpret_cast<void **>(&wide), n1 * sizeof(DWide)));
  BVH_TRY(hipMalloc(reinterpret_cast<void **>(&leaves), n1 * sizeof(DLeaf)));
  if (n == 0) {  // an empty scene: one wide node whose four slots are empty
    DWide w;
    std::memset(&w, 0, sizeof w);
    for (int c = 0; c < 4; ++c) {
      for (int k = 0; k < 3; ++k) {
        w.lo[c][k] = INFINITY;
        w.hi[c][k] = -INFINITY;
      }
      w.child[c] = WIDE_NONE;
    }
    BVH_TRY(hipMemcpy(wide, &w, sizeof w, hipMemcpyHostToDevice));
    BVH_TRY(hipMemset(leaves, 0, sizeof(DLeaf)));
    n_wide = 1;
    n_leaves = 0;
    wide_depth = 0;
    build_ms = ms_since(t0);
    return 0;
  }

  // one block for everything else; the arrays are laid out in the order of BuildArgs
  const size_t sizes[] = {
      6 * n1 * sizeof(double),                              // pbox
      8 * sizeof(unsigned long long),                       // bounds (6 words), counters (2 x 32 bits)
      n1 * sizeof(uint64_t), n1 * sizeof(uint64_t),         // keys, keys_alt
      n1 * sizeof(uint32_t), n1 * sizeof(uint32_t),         // ids, ids_alt
      256 * static_cast<size_t>(a_.sort_blocks) * sizeof(uint32_t),  // hist
      2 * inner * sizeof(int32_t),                          // child
      (2 * n1) * sizeof(int32_t),                           // parent
      2 * inner * sizeof(uint32_t),                         // range
      6 * inner * sizeof(double),                           // nbox
      inner * sizeof(uint32_t),                             // visits
      (n1 + 1) * sizeof(int32_t),                           // wide_src
      n1 * sizeof(int32_t),                                 // leaf_src
  };
  size_t total = 0;
  for (size_t s : sizes) total += padded(s);
  BVH_TRY(hipMalloc(&block_, total));
  char *p = static_cast<char *>(block_);
  auto take = [&](size_t i) {
    char *at = p;
    p += padded(sizes[i]);
    return at;
  };
  a_.pbox = reinterpret_cast<double *>(take(0));
  char *words = take(1);
  a_.bounds = reinterpret_cast<unsigned long long *>(words);
  a_.counters = reinterpret_cast<uint32_t *>(words + 6 * sizeof(unsigned long long));
  a_.keys = reinterpret_cast<uint64_t *>(take(2));
  a_.keys_alt = reinterpret_cast<uint64_t *>(take(3));
  a_.ids = reinterpret_cast<uint32_t *>(take(4));
  a_.ids_alt = reinterpret_cast<uint32_t *>(take(5));
  a_.hist = reinterpret_cast<uint32_t *>(take(6));
  a_.child = reinterpret_cast<int32_t *>(take(7));
  a_.parent = reinterpret_cast<int32_t *>(take(8));
  a_.range = reinterpret_cast<uint32_t *>(take(9));
  a_.nbox = reinterpret_cast<double *>(take(10));
  a_.visits = reinterpret_cast<uint32_t *>(take(11));
  a_.wide_src = reinterpret_cast<int32_t *>(take(12));
  a_.leaf_src = reinterpret_cast<int32_t *>(take(13));
  a_.wide = wide;
  a_.leaves = leaves;

  // bounds start at (+max, -max) in the order-preserving encoding; one wide node (the root) is allocated
  unsigned long long init[8];
  std::memset(init, 0xff, 3 * sizeof(unsigned long long));
  std::memset(init + 3, 0, 3 * sizeof(unsigned long long));
  const uint32_t counters0[2] = {1u, 0u};
  std::memcpy(init + 6, counters0, sizeof counters0);
  BVH_TRY(hipMemcpyAsync(words, init, sizeof init, hipMemcpyHostToDevice, stream));
  BVH_TRY(hipStreamSynchronize(stream));  // `init` is on this stack

  launch_build_boxes(stream, a_);
  launch_build_keys(stream, a_);
  for (int pass = 0; pass < 8; ++pass) {  // 63 bits of code
    a_.shift = 8u * static_cast<uint32_t>(pass);
    launch_build_sort_pass(stream, a_);
    std::swap(a_.keys, a_.keys_alt);
    std::swap(a_.ids, a_.ids_alt);
  }
  BVH_TRY(hipGetLastError());
  int32_t root = ~0;  // a single primitive: the tree is that leaf
  if (n > 1) {
    launch_build_tree(stream, a_);
    BVH_TRY(hipMemsetAsync(a_.visits, 0, inner * sizeof(uint32_t), stream));
    launch_build_fit(stream, a_);
    BVH_TRY(hipGetLastError());
    root = 0;
  }
  BVH_TRY(hipMemcpyAsync(a_.wide_src, &root, sizeof root, hipMemcpyHostToDevice, stream));
  BVH_TRY(hipStreamSynchronize(stream));

  // top-down, a launch per level of the wide tree; the level's size comes back from the device
  uint32_t begin = 0, end = 1;
  int levels = 0;
  while (begin < end) {
    if (++levels > 96) {
      error = "device BVH deeper than 96 wide levels";
      return static_cast<int>(hipErrorUnknown);
    }
    a_.level_begin = begin;
    a_.level_end = end;
    launch_build_emit(stream, a_);
    BVH_TRY(hipGetLastError());
    uint32_t counters[2] = {0, 0};
    BVH_TRY(hipMemcpyAsync(counters, a_.counters, sizeof counters, hipMemcpyDeviceToHost, stream));
    BVH_TRY(hipStreamSynchronize(stream));
    begin = end;
    end = counters[0];
    n_leaves = counters[1];
    if (end > n1 + 1 || n_leaves > n1) {
      error = "device BVH emitted more nodes than primitives";
      return static_cast<int>(hipErrorUnknown);
    }
  }
  n_wide = end;
  wide_depth = levels - 1;
  build_ms = ms_since(t0);
  return 0;
}

int DeviceBvh::refit(hipStream_t stream, const double *d_motion_ends) {
  const auto t0 = std::chrono::steady_clock::now();
  (void)hipGetLastError();
  if (a_.n == 0) return 0;
  if (!block_) {
    error = "no device-built tree to refit";
    return static_cast<int>(hipErrorInvalidValue);
  }
  a_.motion_ends = d_motion_ends;
  launch_build_boxes(stream, a_);
  if (a_.n > 1) {
    BVH_TRY(hipMemsetAsync(a_.visits, 0, (a_.n - 1) * sizeof(uint32_t), stream));
    launch_build_fit(stream, a_);
  }
  launch_build_refresh(stream, a_, n_wide, n_leaves);
  BVH_TRY(hipGetLastError());
  BVH_TRY(hipStreamSynchronize(stream));
  refit_ms = ms_since(t0);
  return 0;
}
