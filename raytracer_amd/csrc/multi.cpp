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
  comm_t comm = nullptr;
  // per frame in flight (slot): a stream, this member's tiles on its device, and "tiles rendered"
  hipStream_t stream[MIRT_GROUP_SLOTS] = {};
  double *tiles[MIRT_GROUP_SLOTS] = {};
  size_t tiles_bytes[MIRT_GROUP_SLOTS] = {};
  hipEvent_t rendered[MIRT_GROUP_SLOTS] = {};
};

struct Pending {
  bool active = false;
  int width = 0, height = 0;
  std::vector<size_t> off;  // in doubles, within `gathered`: where each member's tiles start
};

}  // namespace

struct mirt_group {
  std::vector<Member> m;
  Rccl rccl;
  bool use_rccl = false;
  std::string transport;
  // on the root (member 0) device, per slot: every member's tiles side by side, and the frame
  double *gathered[MIRT_GROUP_SLOTS] = {};
  size_t gathered_bytes[MIRT_GROUP_SLOTS] = {};
  double *frame[MIRT_GROUP_SLOTS] = {};
  size_t frame_bytes[MIRT_GROUP_SLOTS] = {};
  Pending pending[MIRT_GROUP_SLOTS];
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

// Enqueues, behind the members' frames of `slot`: their tiles to the root device, then the kernel that
// puts the tiles where they belong in the frame.  Nothing here waits for a device.
int enqueue_gather(mirt_group *g, int slot) {
  const int n = static_cast<int>(g->m.size());
  const Pending &pd = g->pending[slot];
  Member &root = g->m[0];
  if (g->use_rccl) {
    // one group: the root posts a receive per member, every member posts its send on the stream its
    // frame runs on, so each send starts when that member's tiles are complete
    int rc = g->rccl.GroupStart();
    for (int i = 0; i < n && rc == 0; ++i) {
      const size_t count = pd.off[i + 1] - pd.off[i];
      if (!count) continue;
      rc = g->rccl.Recv(g->gathered[slot] + pd.off[i], count, kNcclDouble, i, root.comm, root.stream[slot]);
      if (rc == 0) rc = g->rccl.Send(g->m[i].tiles[slot], count, kNcclDouble, 0, g->m[i].comm, g->m[i].stream[slot]);
    }
    const int rc_end = g->rccl.GroupEnd();
    if (rc == 0) rc = rc_end;
    if (rc != 0) return mirt_fail(MIRT_ERR_DEVICE, "RCCL gather: %s", g->rccl.GetErrorString(rc));
  } else {
    for (int i = 0; i < n; ++i) {
      const size_t bytes = (pd.off[i + 1] - pd.off[i]) * sizeof(double);
      if (!bytes) continue;
      if (i > 0) {  // the root's stream waits for member i's frame (member 0's frame is on that stream already)
        HIP_TRY(hipSetDevice(g->m[i].device));
        HIP_TRY(hipEventRecord(g->m[i].rendered[slot], g->m[i].stream[slot]));
        HIP_TRY(hipSetDevice(root.device));
        HIP_TRY(hipStreamWaitEvent(root.stream[slot], g->m[i].rendered[slot], 0));
      }
      HIP_TRY(hipSetDevice(root.device));
      HIP_TRY(hipMemcpyPeerAsync(g->gathered[slot] + pd.off[i], root.device, g->m[i].tiles[slot], g->m[i].device, bytes,
                                 root.stream[slot]));
    }
  }
  HIP_TRY(hipSetDevice(root.device));
  DeinterleaveArgs da;
  std::memset(&da, 0, sizeof da);
  da.gathered = g->gathered[slot];
  da.frame = g->frame[slot];
  da.width = pd.width;
  da.height = pd.height;
  da.tile = MIRT_GROUP_TILE;
  da.tiles_x = (pd.width + MIRT_GROUP_TILE - 1) / MIRT_GROUP_TILE;
  da.world = n;
  da.stagger = tile_stagger(da.tiles_x, n);
  for (int i = 0; i < n; ++i) da.offset[i] = pd.off[i];
  launch_deinterleave(static_cast<unsigned>((static_cast<size_t>(pd.width) * pd.height + 255) / 256), root.stream[slot], da);
  HIP_TRY(hipGetLastError());
  return MIRT_OK;
}

// Waits for whatever the group has enqueued for `slot` (after an error: nothing may still be running
// when the caller is told).
void drain(mirt_group *g, int slot) {
  for (Member &mb : g->m) {
    if (hipSetDevice(mb.device) == hipSuccess && mb.stream[slot]) (void)hipStreamSynchronize(mb.stream[slot]);
  }
}

}  // namespace

MIRT_API void mirt_group_destroy(mirt_group *g) {
  if (!g) return;
  for (Member &mb : g->m) {
    (void)hipSetDevice(mb.device);
    if (mb.comm && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(mb.comm);
    for (int s = 0; s < MIRT_GROUP_SLOTS; ++s) {
      if (mb.tiles[s]) (void)hipFree(mb.tiles[s]);
      if (mb.stream[s]) (void)hipStreamDestroy(mb.stream[s]);
      if (mb.rendered[s]) (void)hipEventDestroy(mb.rendered[s]);
    }
    mirt_scene_destroy(mb.scene);
  }
  if (!g->m.empty()) {
    (void)hipSetDevice(g->m[0].device);
    for (int s = 0; s < MIRT_GROUP_SLOTS; ++s) {
      if (g->gathered[s]) (void)hipFree(g->gathered[s]);
      if (g->frame[s]) (void)hipFree(g->frame[s]);
    }
  }
  if (g->rccl.lib) (void)dlclose(g->rccl.lib);
  delete g;
}

MIRT_API int mirt_group_create(const mirt_scene_desc *desc, const int *devices, int n_devices, mirt_group **out) {
  return mirt_group_create_with(desc, devices, n_devices, nullptr, out);
}

MIRT_API int mirt_group_create_with(const mirt_scene_desc *desc, const int *devices, int n_devices,
                                    const mirt_build_options *options, mirt_group **out) {
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
    if (int rc = mirt_scene_create_with(desc, devices[i], options, &mb.scene)) return rc;  // a replica per member
    HIP_TRY(hipSetDevice(mb.device));
    for (int s = 0; s < MIRT_GROUP_SLOTS; ++s) {
      HIP_TRY(hipStreamCreateWithFlags(&mb.stream[s], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&mb.rendered[s], hipEventDisableTiming));
    }
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

MIRT_API int mirt_group_render_begin(mirt_group *g, const mirt_camera *camera, const mirt_config *config, int width,
                                     int height, int spp, int slot) {
  if (!g || !camera || !config) return mirt_fail(MIRT_ERR_INVALID, "NULL argument");
  if (width <= 0 || height <= 0 || spp <= 0) return mirt_fail(MIRT_ERR_INVALID, "width, height and spp must be positive");
  if (slot < 0 || slot >= MIRT_GROUP_SLOTS) return mirt_fail(MIRT_ERR_INVALID, "slot must be 0 .. %d", MIRT_GROUP_SLOTS - 1);
  Pending &pd = g->pending[slot];
  if (pd.active) return mirt_fail(MIRT_ERR_INVALID, "slot %d still holds a frame: call mirt_group_render_end first", slot);
  const int n = static_cast<int>(g->m.size());
  const int tile = MIRT_GROUP_TILE;
  Member &root = g->m[0];
  try {
    pd.off.assign(n + 1, 0);
  } catch (const std::exception &) {
    return mirt_fail(MIRT_ERR_MEMORY, "out of host memory");
  }
  for (int i = 0; i < n; ++i) {
    Member &mb = g->m[i];
    const size_t px = static_cast<size_t>(mirt_tile_buffer_pixels(width, height, tile, tile, i, n));
    pd.off[i + 1] = pd.off[i] + px * 3;
    if (int rc = grow(mb.device, &mb.tiles[slot], &mb.tiles_bytes[slot], std::max<size_t>(px * 3 * sizeof(double), 64)))
      return rc;
  }
  if (int rc = grow(root.device, &g->gathered[slot], &g->gathered_bytes[slot], std::max<size_t>(pd.off[n] * sizeof(double), 64)))
    return rc;
  const size_t frame_bytes = static_cast<size_t>(width) * height * 3 * sizeof(double);
  if (int rc = grow(root.device, &g->frame[slot], &g->frame_bytes[slot], frame_bytes)) return rc;
  pd.width = width;
  pd.height = height;
  // every member's share, enqueued on its own device and stream: they render at the same time
  const char *hook = std::getenv("MIRT_GROUP_TEST_FAIL_MEMBER");  // test hook: this member's begin fails
  const int fail_member = hook && *hook ? std::atoi(hook) : -1;
  for (int i = 0; i < n; ++i) {
    Member &mb = g->m[i];
    int rc = i == fail_member ? mirt_fail(MIRT_ERR_DEVICE, "member %d: failure injected by MIRT_GROUP_TEST_FAIL_MEMBER", i)
                              : mirt_render_tiles_begin(mb.scene, camera, config, width, height, spp, tile, tile, i, n,
                                                        mb.tiles[slot], mb.stream[slot], slot);
    if (rc != MIRT_OK) {
      // the members already begun hold a frame in this slot: finish them, or the slot stays taken for good
      const std::string why = mirt_last_error();
      for (int j = 0; j < i; ++j) (void)mirt_render_tiles_end(g->m[j].scene, slot, nullptr);
      return mirt_fail(rc, "%s", why.c_str());
    }
  }
  if (int rc = enqueue_gather(g, slot)) {
    const std::string why = mirt_last_error();
    for (int j = 0; j < n; ++j) (void)mirt_render_tiles_end(g->m[j].scene, slot, nullptr);
    drain(g, slot);
    return mirt_fail(rc, "%s", why.c_str());
  }
  pd.active = true;
  return MIRT_OK;
}

MIRT_API int mirt_group_render_end(mirt_group *g, int slot, double *rgb_out, mirt_stats *stats, mirt_stats *member_stats) {
  if (!g) return mirt_fail(MIRT_ERR_INVALID, "NULL group");
  if (slot < 0 || slot >= MIRT_GROUP_SLOTS) return mirt_fail(MIRT_ERR_INVALID, "slot must be 0 .. %d", MIRT_GROUP_SLOTS - 1);
  Pending &pd = g->pending[slot];
  if (!pd.active) return mirt_fail(MIRT_ERR_INVALID, "no frame in flight in slot %d", slot);
  pd.active = false;
  const int n = static_cast<int>(g->m.size());
  Member &root = g->m[0];
  // wait for every member's frame (a member whose queues overflowed renders it again here); every member
  // is waited for even after one has failed, so that no slot stays taken
  mirt_stats total;
  std::memset(&total, 0, sizeof total);
  int first_rc = MIRT_OK;
  std::string why;
  bool redone = false;
  for (int i = 0; i < n; ++i) {
    mirt_stats st;
    std::memset(&st, 0, sizeof st);
    const int rc = mirt_render_tiles_end(g->m[i].scene, slot, &st);
    if (rc != MIRT_OK) {
      if (first_rc == MIRT_OK) {
        first_rc = rc;
        why = mirt_last_error();
      }
      continue;
    }
    if (member_stats) member_stats[i] = st;
    redone = redone || st.frame_attempts > 1;
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
  if (first_rc != MIRT_OK) {
    drain(g, slot);
    return mirt_fail(first_rc, "%s", why.c_str());
  }
  // a frame that was rendered again was gathered half-done the first time
  if (redone)
    if (int rc = enqueue_gather(g, slot)) {
      drain(g, slot);
      return rc;
    }
  HIP_TRY(hipSetDevice(root.device));
  HIP_TRY(hipStreamSynchronize(root.stream[slot]));
  if (g->use_rccl)
    for (int i = 1; i < n; ++i) {  // the senders' streams (RCCL enqueued the sends there)
      HIP_TRY(hipSetDevice(g->m[i].device));
      HIP_TRY(hipStreamSynchronize(g->m[i].stream[slot]));
    }
  if (rgb_out) {
    HIP_TRY(hipSetDevice(root.device));
    if (int rc = mirt_copy_frame_out(root.scene, rgb_out, g->frame[slot],
                                     static_cast<size_t>(pd.width) * pd.height * 3 * sizeof(double)))
      return rc;
  }
  if (stats) *stats = total;
  return MIRT_OK;
}

MIRT_API const double *mirt_group_frame(const mirt_group *g, int slot) {
  if (!g || slot < 0 || slot >= MIRT_GROUP_SLOTS) return nullptr;
  return g->frame[slot];
}

MIRT_API int mirt_group_render(mirt_group *g, const mirt_camera *camera, const mirt_config *config, int width,
                               int height, int spp, double *rgb_out, mirt_stats *stats, mirt_stats *member_stats) {
  if (!rgb_out) return mirt_fail(MIRT_ERR_INVALID, "NULL argument");
  if (int rc = mirt_group_render_begin(g, camera, config, width, height, spp, 0)) return rc;
  return mirt_group_render_end(g, 0, rgb_out, stats, member_stats);
}
