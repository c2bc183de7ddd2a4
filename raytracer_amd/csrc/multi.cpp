// Several GPUs behind the C ABI, one process: mirt_group_* of include/mirt.h.
//
// The reference has one address space and one OpenMP loop (Code/Render/Raytracer.cpp:525); the
// engine's counterpart of "more threads" is more GPUs: the scene is replicated (one mirt_scene per
// device), the image is cut into 16x16-pixel tiles dealt round-robin (tile t to member t % n), every
// member renders its tiles into a compact buffer on its own device and stream at the same time, and
// the one exchange of the frame -- member buffers to the first member's device -- goes over RCCL
// (ncclSend / ncclRecv in one group: every sender uses its own xGMI link to the root at once).  RCCL
// is loaded at run time (librccl.so); where it is missing, or refuses the device list (the same
// device named twice, as the single-GPU rehearsal of this code does), the buffers are moved by
// hipMemcpyPeerAsync instead and mirt_group_transport() says so.  A kernel on the root device then
// puts the tiles where they belong in the frame.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "device_types.h"
#include "mirt.h"

#define MIRT_API extern "C" __attribute__((visibility("default")))

int mirt_fail(int code, const char *fmt, ...);  // engine.cpp: sets mirt_last_error()
int mirt_copy_frame_out(mirt_scene *sc, void *dst, const void *src, size_t bytes);  // engine.cpp: staged read-back

namespace {

// the part of RCCL this file uses (rccl.h: ncclCommInitAll, ncclSend, ncclRecv, ncclGroupStart/End)
typedef void *comm_t;
struct Rccl {
  void *lib = nullptr;
  int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool load() {
    if (lib) return true;
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return false;
    auto sym = [&](const char *n) { return dlsym(lib, n); };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    return CommInitAll && CommDestroy && Send && Recv && GroupStart && GroupEnd && GetErrorString;
  }
};
constexpr int kNcclDouble = 8;  // ncclFloat64 (rccl.h)

#define HIP_TRY(expr)                                                                                    \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess)                                                                                \
      return mirt_fail(e_ == hipErrorOutOfMemory ? MIRT_ERR_MEMORY : MIRT_ERR_DEVICE, "%s: %s", #expr,  \
                       hipGetErrorString(e_));                                                           \
  } while (0)

struct Member {
  int device = 0;
  mirt_scene *scene = nullptr;
  hipStream_t stream = nullptr;
  double *tiles = nullptr;  // this member's tiles, on its device
  size_t tiles_bytes = 0;
  comm_t comm = nullptr;
};

}  // namespace

struct mirt_group {
  std::vector<Member> m;
  Rccl rccl;
  bool use_rccl = false;
  std::string transport;
  // on the root (member 0) device: every member's tiles side by side, and the frame
  double *gathered = nullptr;
  size_t gathered_bytes = 0;
  double *frame = nullptr;
  size_t frame_bytes = 0;
  hipEvent_t done[MIRT_GROUP_MAX] = {};
};

namespace {

int grow(int device, double **p, size_t *have, size_t want) {
  if (want <= *have) return MIRT_OK;
  HIP_TRY(hipSetDevice(device));
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *have = 0;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), want));
  *have = want;
  return MIRT_OK;
}

}  // namespace

MIRT_API void mirt_group_destroy(mirt_group *g) {
  if (!g) return;
  for (Member &mb : g->m) {
    (void)hipSetDevice(mb.device);
    if (mb.comm && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(mb.comm);
    if (mb.tiles) (void)hipFree(mb.tiles);
    if (mb.stream) (void)hipStreamDestroy(mb.stream);
    mirt_scene_destroy(mb.scene);
  }
  if (!g->m.empty()) {
    (void)hipSetDevice(g->m[0].device);
    if (g->gathered) (void)hipFree(g->gathered);
    if (g->frame) (void)hipFree(g->frame);
  }
  for (hipEvent_t e : g->done)
    if (e) (void)hipEventDestroy(e);
  delete g;
}

MIRT_API int mirt_group_create(const mirt_scene_desc *desc, const int *devices, int n_devices, mirt_group **out) {
  if (!out) return mirt_fail(MIRT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > MIRT_GROUP_MAX)
    return mirt_fail(MIRT_ERR_INVALID, "a group has 1 to %d members", MIRT_GROUP_MAX);
  mirt_group *g = nullptr;
  try {
    g = new mirt_group;
  } catch (const std::exception &) {
    return mirt_fail(MIRT_ERR_MEMORY, "out of host memory");
  }
  struct Guard {
    mirt_group *g;
    ~Guard() { mirt_group_destroy(g); }
  } guard{g};
  g->m.resize(n_devices);
  for (int i = 0; i < n_devices; ++i) {
    Member &mb = g->m[i];
    mb.device = devices[i];
    if (int rc = mirt_scene_create(desc, devices[i], &mb.scene)) return rc;  // a replica per member
    HIP_TRY(hipSetDevice(mb.device));
    HIP_TRY(hipStreamCreateWithFlags(&mb.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&g->done[i], hipEventDisableTiming));
  }
  // RCCL: one communicator per member, all made by this process.  It needs distinct devices.
  bool distinct = true;
  for (int i = 0; i < n_devices; ++i)
    for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
  const char *force = std::getenv("MIRT_GROUP_TRANSPORT");  // "copy": do not use RCCL
  if (force && std::strcmp(force, "copy") == 0) {
    g->transport = "hipMemcpyPeerAsync (MIRT_GROUP_TRANSPORT=copy)";
  } else if (!distinct) {
    g->transport = "hipMemcpyPeerAsync (a device is named twice: RCCL wants one rank per device)";
  } else if (!g->rccl.load()) {
    g->transport = "hipMemcpyPeerAsync (librccl.so could not be loaded)";
  } else {
    std::vector<comm_t> comms(n_devices, nullptr);
    const int rc = g->rccl.CommInitAll(comms.data(), n_devices, devices);
    if (rc != 0) {
      g->transport = std::string("hipMemcpyPeerAsync (ncclCommInitAll: ") + g->rccl.GetErrorString(rc) + ")";
    } else {
      for (int i = 0; i < n_devices; ++i) g->m[i].comm = comms[i];
      g->use_rccl = true;
      g->transport = "RCCL ncclSend/ncclRecv";
    }
  }
  guard.g = nullptr;
  *out = g;
  return MIRT_OK;
}

MIRT_API const char *mirt_group_transport(const mirt_group *g) { return g ? g->transport.c_str() : ""; }
MIRT_API int mirt_group_size(const mirt_group *g) { return g ? static_cast<int>(g->m.size()) : 0; }

MIRT_API int mirt_group_set_traversal(mirt_group *g, int mode) {
  if (!g) return mirt_fail(MIRT_ERR_INVALID, "NULL group");
  for (Member &mb : g->m)
    if (int rc = mirt_scene_set_traversal(mb.scene, mode)) return rc;
  return MIRT_OK;
}

MIRT_API int mirt_group_render(mirt_group *g, const mirt_camera *camera, const mirt_config *config, int width,
                               int height, int spp, double *rgb_out, mirt_stats *stats, mirt_stats *member_stats) {
  if (!g || !camera || !config || !rgb_out) return mirt_fail(MIRT_ERR_INVALID, "NULL argument");
  if (width <= 0 || height <= 0 || spp <= 0) return mirt_fail(MIRT_ERR_INVALID, "width, height and spp must be positive");
  const int n = static_cast<int>(g->m.size());
  const int tile = MIRT_GROUP_TILE;
  Member &root = g->m[0];

  // every member's share, enqueued on its own device and stream: they render at the same time
  std::vector<size_t> off(n + 1, 0);  // in doubles, within `gathered`
  for (int i = 0; i < n; ++i) {
    Member &mb = g->m[i];
    const size_t px = static_cast<size_t>(mirt_tile_buffer_pixels(width, height, tile, tile, i, n));
    off[i + 1] = off[i] + px * 3;
    if (int rc = grow(mb.device, &mb.tiles, &mb.tiles_bytes, std::max<size_t>(px * 3 * sizeof(double), 64))) return rc;
  }
  if (int rc = grow(root.device, &g->gathered, &g->gathered_bytes, std::max<size_t>(off[n] * sizeof(double), 64))) return rc;
  const size_t frame_bytes = static_cast<size_t>(width) * height * 3 * sizeof(double);
  if (int rc = grow(root.device, &g->frame, &g->frame_bytes, frame_bytes)) return rc;
  for (int i = 0; i < n; ++i) {
    Member &mb = g->m[i];
    if (int rc = mirt_render_tiles_begin(mb.scene, camera, config, width, height, spp, tile, tile, i, n, mb.tiles,
                                         mb.stream, 0))
      return rc;
  }
  // wait for every member's frame (a member whose queues overflowed renders it again here), then move
  mirt_stats total;
  std::memset(&total, 0, sizeof total);
  for (int i = 0; i < n; ++i) {
    mirt_stats st;
    if (int rc = mirt_render_tiles_end(g->m[i].scene, 0, &st)) return rc;
    if (member_stats) member_stats[i] = st;
    total.primary_rays += st.primary_rays;
    total.secondary_rays += st.secondary_rays;
    total.shadow_segments += st.shadow_segments;
    total.node_tests += st.node_tests;
    total.prim_tests += st.prim_tests;
    total.shadow_node_tests += st.shadow_node_tests;
    total.shadow_prim_tests += st.shadow_prim_tests;
    total.launches += st.launches;
    total.frame_attempts = std::max(total.frame_attempts, st.frame_attempts);
    total.device_ms = std::max(total.device_ms, st.device_ms);  // the members run side by side
    total.trace_kernel_ms += st.trace_kernel_ms;
    total.shadow_kernel_ms += st.shadow_kernel_ms;
    total.trace_kernel_launches += st.trace_kernel_launches;
    total.shadow_kernel_launches += st.shadow_kernel_launches;
  }
  if (g->use_rccl) {
    // one group: the root posts a receive per member, every member posts its send
    int rc = g->rccl.GroupStart();
    for (int i = 0; i < n && rc == 0; ++i) {
      const size_t count = off[i + 1] - off[i];
      if (!count) continue;
      rc = g->rccl.Recv(g->gathered + off[i], count, kNcclDouble, i, root.comm, root.stream);
      if (rc == 0) rc = g->rccl.Send(g->m[i].tiles, count, kNcclDouble, 0, g->m[i].comm, g->m[i].stream);
    }
    const int rc_end = g->rccl.GroupEnd();
    if (rc == 0) rc = rc_end;
    if (rc != 0) return mirt_fail(MIRT_ERR_DEVICE, "RCCL gather: %s", g->rccl.GetErrorString(rc));
  } else {
    for (int i = 0; i < n; ++i) {
      const size_t bytes = (off[i + 1] - off[i]) * sizeof(double);
      if (!bytes) continue;
      HIP_TRY(hipSetDevice(root.device));
      HIP_TRY(hipMemcpyPeerAsync(g->gathered + off[i], root.device, g->m[i].tiles, g->m[i].device, bytes, root.stream));
    }
  }
  // tiles -> frame on the root device, then to the caller's buffer
  HIP_TRY(hipSetDevice(root.device));
  DeinterleaveArgs da;
  std::memset(&da, 0, sizeof da);
  da.gathered = g->gathered;
  da.frame = g->frame;
  da.width = width;
  da.height = height;
  da.tile = tile;
  da.tiles_x = (width + tile - 1) / tile;
  da.world = n;
  for (int i = 0; i < n; ++i) da.offset[i] = off[i];
  launch_deinterleave(static_cast<unsigned>((static_cast<size_t>(width) * height + 255) / 256), root.stream, da);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(root.stream));
  if (int rc = mirt_copy_frame_out(root.scene, rgb_out, g->frame, frame_bytes)) return rc;
  for (int i = 1; i < n; ++i) {  // the senders' streams (RCCL enqueued the sends there)
    HIP_TRY(hipSetDevice(g->m[i].device));
    HIP_TRY(hipStreamSynchronize(g->m[i].stream));
  }
  if (stats) *stats = total;
  return MIRT_OK;
}
