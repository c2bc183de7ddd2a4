// Host side of the engine: the C ABI of include/mirt.h.  Derives the device layout from a
// mirt_scene_desc, owns the device memory, and drives the per-depth kernel pipeline of pipeline.h.
// Compiled as plain C++ (no FMA contraction: the matrices prepared here are the same numbers the
// reference computes); the kernels are reached through the launchers of kernels.hip.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../host/mirt_math.h"
#include "bvh_device.h"
#include "device_types.h"
#include "mirt.h"
#include "mirt_rng.h"

#define MIRT_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_error;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(e_ == hipErrorOutOfMemory ? MIRT_ERR_MEMORY : MIRT_ERR_DEVICE, "%s: %s", #expr, \
                  hipGetErrorString(e_));                                                          \
  } while (0)

struct DeviceBuffer {
  void *p = nullptr;
  size_t bytes = 0;
  bool owned = true;  // false: a window of memory that belongs to somebody else (never freed, never grown)
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  DeviceBuffer(DeviceBuffer &&o) noexcept : p(o.p), bytes(o.bytes), owned(o.owned) {
    o.p = nullptr;
    o.bytes = 0;
  }
  ~DeviceBuffer() { release(); }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  void borrow(void *mem, size_t n) {
    release();
    owned = false;
    p = mem;
    bytes = n;
  }
  hipError_t reserve(size_t n) {  // grow-only
    if (n <= bytes) return hipSuccess;
    if (!owned) return hipErrorOutOfMemory;
    release();
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <class T> T *as() const { return static_cast<T *>(p); }
};

template <class T> hipError_t upload(DeviceBuffer &b, const std::vector<T> &v) {
  // an empty array still gets one zeroed record: kernels bind references to element 0 before they
  // know whether a lane has a hit, and a reference may be loaded from ahead of its guard
  hipError_t e = b.reserve(std::max<size_t>(v.size() * sizeof(T), std::max<size_t>(sizeof(T), 16)));
  if (e != hipSuccess) return e;
  if (v.empty()) return hipMemset(b.p, 0, b.bytes);
  return hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

struct RayQueueMem {
  DeviceBuffer mem;
  DRayQueue q{};
  RayQueueMem() = default;
  RayQueueMem(RayQueueMem &&o) noexcept : q(o.q) {
    mem.p = o.mem.p;
    mem.bytes = o.mem.bytes;
    mem.owned = o.mem.owned;
    o.mem.p = nullptr;
    o.mem.bytes = 0;
    o.q = DRayQueue{};
  }
  RayQueueMem(const RayQueueMem &) = delete;
  RayQueueMem &operator=(const RayQueueMem &) = delete;
};

struct Timed {
  hipEvent_t a, b;
  int kind;  // 0 trace, 1 shadow
};

// which kinds of ray-spawning materials the scene has (decides the most children one hit can have)
struct SpawnKinds {
  bool mirror = false, lobe = false, glass = false, mirror_glass = false, lobe_glass = false;
};

}  // namespace

// for the other translation units of the library: records the message of mirt_last_error()
int mirt_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return fail(code, "%s", buf);
}

struct mirt_scene {
  int device = 0;
  int walk = WALK_PACKET;  // BVH walk (WALK_*)
  int n_prims = 0;
  int bvh_depth = 0;  // edges from the root to the deepest node
  hipStream_t stream = nullptr;
  hipEvent_t null_stream_mark = nullptr;  // orders the engine's own stream after the caller's null stream
  DScene d{};
  int wide_depth = 0;  // levels of the wide walk's tree below its root
  uint32_t n_wide = 0, n_leaves = 0;  // of the tree collapsed on the host
  unsigned resident_grid = 768;  // blocks of a launch whose waves stay and take work off a queue: 3 per CU
  bool stream_kernels = true;
  DeviceBuffer nodes, leaf_prims, leaf_tagged, wide, leaves, prims, motion, materials, transparency, lights, textures, light_frame;
  // MIRT_BUILDER_DEVICE: the tree was built on the GPU (the wide walk's arrays belong to `dbvh`), and what
  // mirt_scene_update_prims / mirt_scene_refit need to move its primitives afterwards
  int builder = MIRT_BUILDER_HOST;
  std::unique_ptr<DeviceBvh> dbvh;
  DeviceBuffer motion_ends;                 // 24 doubles per moving primitive: rows 0-2 of matrix_t0 and of matrix_t1
  std::vector<double> motion_ends_host;
  std::vector<DMotion> motion_host;
  std::vector<int32_t> prim_material, prim_motion_index;  // per primitive; motion index -1: static
  std::vector<uint8_t> prim_type;
  double create_ms = 0.0, tree_ms = 0.0;    // wall clock of mirt_scene_create, and of its BVH part
  std::vector<DeviceBuffer> texels;
  std::vector<mirt_light> host_lights;
  SpawnKinds kinds;
  // Per-frame working memory, grown on demand and kept.  Two sets, so that a second frame can be
  // enqueued (mirt_render_tiles_begin) while the first is still running.
  struct WorkSet {
    bool fixed = false;  // working memory is a carve-up of the per-GPU static array: nothing is ever allocated
    RayQueueMem queues[2];  // rays of even / odd depth: a depth reads one and appends to the other
    DeviceBuffer shadow_items, control, accum;  // control: statistics counters, then the queue tails
    DeviceBuffer hits, hits_uv;  // closest hits between k_closest_stream and k_shade (large scenes only)
    char *pinned = nullptr;  // read-back of the control block
    size_t pinned_bytes = 0;
    size_t ray_shard_cap = 0, shadow_shard_cap = 0;  // entries per queue shard
    bool accum_clean = false;  // the accumulation buffer is all zero (k_resolve leaves it so)
    size_t accum_clean_bytes = 0;
    std::vector<hipEvent_t> events;
    // the frame in flight
    bool pending = false;
    mirt_camera camera{};
    mirt_config config{};
    int width = 0, height = 0, spp = 0, tile_w = 0, tile_h = 0, rank = 0, world = 1;
    double *d_out = nullptr;
    hipStream_t stream = nullptr;
    size_t chunks = 0, control_bytes = 0;
    uint64_t per_launch = 0, launches = 0, attempts = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<Timed> timed;
    ~WorkSet() {
      for (auto e : events) (void)hipEventDestroy(e);
      if (pinned) (void)hipHostFree(pinned);
    }
  } work[MIRT_SLOTS], fixed_work;
  char *scratch = nullptr;  // device address of this GPU's static array (scratch_address)
  std::vector<DLightFrame> light_frame_host;
  DeviceBuffer out, ids;
  // progress feed (mirt_scene_record_progress): wall clock per band of 16 pixel rows, and the last frame's read-back
  bool record_progress = false;
  DeviceBuffer progress;
  std::vector<unsigned long long> progress_host;
  int progress_bands = 0;
  double wall_clock_khz = 100000.0;
  uint64_t split_for = 0;  // frame shape (primary grid blocks, depth, glossy samples) `split` was found for
  uint32_t split = 1;      // primary launches are cut this many times finer than the queues' sizes suggest
  int light_frame_for = 0;  // shadow_samples the device copy of the lights' strata was made for
  bool light_frame_valid = false;
  bool any_strata = false;  // some light of the current light-frame table has more than one stratum
  bool point_lights = false;  // no light of the scene is an area light
  // frame to a caller's host buffer: pinned staging memory and one event per piece (copy_frame_out)
  static constexpr int kStagePieces = 4;
  char *stage = nullptr;
  size_t stage_bytes = 0;
  hipEvent_t stage_done[kStagePieces] = {};

  ~mirt_scene() {
    if (stage) (void)hipHostFree(stage);
    for (hipEvent_t e : stage_done)
      if (e) (void)hipEventDestroy(e);
    if (null_stream_mark) (void)hipEventDestroy(null_stream_mark);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

Mat4 mat_of(const double *a) {
  Mat4 m;
  std::memcpy(m.m, a, sizeof(double) * 16);
  return m;
}

void store_3x4(double *dst, const Mat4 &m) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) dst[r * 4 + c] = m.m[r][c];
}

int validate(const mirt_scene_desc *s, bool device_tree) {
  if (!s) return fail(MIRT_ERR_INVALID, "scene description is NULL");
  if (!device_tree && (s->n_nodes < 1 || !s->nodes)) return fail(MIRT_ERR_INVALID, "scene needs at least one BVH node");
  if (s->n_prims < 0 || s->n_leaf_prims < 0 || s->n_materials < 0 || s->n_lights < 0 || s->n_textures < 0 ||
      s->n_plane_points < 0)
    return fail(MIRT_ERR_INVALID, "negative count in scene description");
  if ((s->n_prims && !s->prims) || (!device_tree && s->n_leaf_prims && !s->leaf_prims) || (s->n_materials && !s->materials) ||
      (s->n_lights && !s->lights) || (s->n_textures && !s->textures) || (s->n_plane_points && !s->plane_points))
    return fail(MIRT_ERR_INVALID, "NULL array with a non-zero count");
  for (int i = 0; i < s->n_nodes && !device_tree; ++i) {  // (a tree built on the device ignores the caller's)
    const mirt_node &n = s->nodes[i];
    if (n.left < 0) {
      if (n.count < 0 || n.first < 0 || (long long)n.first + n.count > s->n_leaf_prims)
        return fail(MIRT_ERR_INVALID, "leaf node %d names leaf_prims[%d..%d) outside [0,%d)", i, n.first,
                    n.first + n.count, s->n_leaf_prims);
    } else if (n.left != i + 1 || n.right <= n.left || n.right >= s->n_nodes) {
      return fail(MIRT_ERR_INVALID, "inner node %d: children (%d,%d) are not in depth-first order", i, n.left,
                  n.right);
    }
  }
  for (int i = 0; i < s->n_leaf_prims && !device_tree; ++i)
    if (s->leaf_prims[i] < 0 || s->leaf_prims[i] >= s->n_prims)
      return fail(MIRT_ERR_INVALID, "leaf_prims[%d] = %d is not a primitive", i, s->leaf_prims[i]);
  for (int i = 0; i < s->n_prims; ++i) {
    const mirt_prim &p = s->prims[i];
    if (p.type < MIRT_SPHERE || p.type > MIRT_CONE) return fail(MIRT_ERR_INVALID, "primitive %d: unknown type %d", i, p.type);
    if (p.material < 0 || p.material >= s->n_materials)
      return fail(MIRT_ERR_INVALID, "primitive %d: material %d out of range", i, p.material);
    if (p.type == MIRT_PLANE && (p.n_points < 0 || p.first_point < 0 ||
                                 (long long)p.first_point + p.n_points > s->n_plane_points))
      return fail(MIRT_ERR_INVALID, "plane %d: points out of range", i);
  }
  for (int i = 0; i < s->n_materials; ++i) {
    const mirt_material &m = s->materials[i];
    const int idx[3] = {m.texture, m.normal_map, m.bump_map};
    for (int k : idx)
      if (k < -1 || k >= s->n_textures) return fail(MIRT_ERR_INVALID, "material %d: texture index %d out of range", i, k);
  }
  for (int i = 0; i < s->n_textures; ++i) {
    const mirt_texture &t = s->textures[i];
    if (t.width <= 0 || t.height <= 0 || t.max_value <= 0 || !t.rgb)
      return fail(MIRT_ERR_INVALID, "texture %d is empty", i);
  }
  return MIRT_OK;
}

// index of the first node after the subtree of node i (nodes are in depth-first order)
int fill_skip(const mirt_scene_desc &s, std::vector<DNode> &out, int i, int depth, int &max_depth) {
  const mirt_node &n = s.nodes[i];
  max_depth = std::max(max_depth, depth);
  DNode &d = out[i];
  for (int a = 0; a < 3; ++a) {
    d.bmin[a] = n.bmin[a];
    d.bmax[a] = n.bmax[a];
  }
  d.order = 0;
  int next;
  if (n.left < 0) {
    d.first = n.first;
    d.count = n.count;
    next = i + 1;
  } else {
    d.first = -1;
    d.count = n.right;  // inner node: index of the second child (the first is i + 1)
    const int after_left = fill_skip(s, out, n.left, depth + 1, max_depth);
    if (after_left != n.right) return -1;
    next = fill_skip(s, out, n.right, depth + 1, max_depth);
    if (next < 0) return -1;
    // for the ordered walk: the axis along which the children's boxes are furthest apart, and
    // whether the second child is the lower one along it
    const mirt_node &l = s.nodes[n.left], &r = s.nodes[n.right];
    int axis = 0;
    double best = -1.0;
    for (int a = 0; a < 3; ++a) {
      const double sep = std::fabs((r.bmin[a] + r.bmax[a]) - (l.bmin[a] + l.bmax[a]));
      if (sep > best) {
        best = sep;
        axis = a;
      }
    }
    const bool second_lower = (r.bmin[axis] + r.bmax[axis]) < (l.bmin[axis] + l.bmax[axis]);
    out[i].order = axis | (second_lower ? 4 : 0);
  }
  out[i].skip = next;
  return next;
}

// ---- the wide walk's view of the tree (device_types.h: DWide, DLeaf) ------------------------------
float round_down(double x) {
  float f = static_cast<float>(x);
  if (static_cast<double>(f) > x) f = std::nextafterf(f, -INFINITY);
  return f;
}
float round_up(double x) {
  float f = static_cast<float>(x);
  if (static_cast<double>(f) < x) f = std::nextafterf(f, INFINITY);
  return f;
}

struct WideBuilder {
  const mirt_scene_desc &s;
  const std::vector<DNode> &nodes;   // for the order codes
  const std::vector<int32_t> &tagged;
  std::vector<DWide> wide;
  std::vector<DLeaf> leaves;
  int depth = 0;  // wide levels below the root

  int32_t leaf_ref(int i) {
    const mirt_node &n = s.nodes[i];
    DLeaf l;
    for (int a = 0; a < 3; ++a) {
      l.bmin[a] = n.bmin[a];
      l.bmax[a] = n.bmax[a];
    }
    for (int k = 0; k < 4; ++k) l.prim[k] = -1;
    if (n.count <= 4) {
      for (int k = 0; k < n.count; ++k) l.prim[k] = tagged[n.first + k];
    } else {
      l.prim[0] = n.first;
      l.prim[1] = n.count;
      l.prim[3] = LEAF_INDIRECT;
    }
    leaves.push_back(l);
    return ~static_cast<int32_t>(leaves.size() - 1);
  }
  void set_slot(DWide &w, int slot, int i) {
    const mirt_node &n = s.nodes[i];
    for (int a = 0; a < 3; ++a) {
      w.lo[slot][a] = round_down(n.bmin[a]);
      w.hi[slot][a] = round_up(n.bmax[a]);
    }
  }
  // the wide node of binary inner node t (or of a root that is a leaf); returns its index
  int32_t build(int t, int level) {
    depth = std::max(depth, level);
    const int32_t me = static_cast<int32_t>(wide.size());
    wide.emplace_back();
    DWide w;
    std::memset(&w, 0, sizeof w);
    for (int c = 0; c < 4; ++c) {
      for (int a = 0; a < 3; ++a) {
        w.lo[c][a] = INFINITY;
        w.hi[c][a] = -INFINITY;
      }
      w.child[c] = WIDE_NONE;
    }
    const mirt_node &n = s.nodes[t];
    if (n.left < 0) {  // the whole tree is one leaf
      set_slot(w, 0, t);
      w.child[0] = leaf_ref(t);
    } else {
      w.order = static_cast<uint32_t>(nodes[t].order) & 7u;
      const int kids[2] = {n.left, n.right};
      for (int g = 0; g < 2; ++g) {
        const int x = kids[g];
        const mirt_node &nx = s.nodes[x];
        if (nx.left < 0) {
          set_slot(w, 2 * g, x);
          w.child[2 * g] = leaf_ref(x);
        } else {
          w.order |= (static_cast<uint32_t>(nodes[x].order) & 7u) << (3 * (g + 1));
          const int gk[2] = {nx.left, nx.right};
          for (int h = 0; h < 2; ++h) {
            set_slot(w, 2 * g + h, gk[h]);
            w.child[2 * g + h] = s.nodes[gk[h]].left < 0 ? leaf_ref(gk[h]) : build(gk[h], level + 1);
          }
        }
      }
    }
    wide[me] = w;
    return me;
  }
};

// Plane.cpp:15-19, :46-56, :106-113 evaluated once.
void prepare_plane(const mirt_scene_desc &s, const mirt_prim &p, DPrim &d) {
  if (p.n_points < 3) {  // Plane.cpp:11-13: never hit
    d.visible = 0;
    return;
  }
  const double *pts = s.plane_points + 3 * (size_t)p.first_point;
  const Vec3 p0(pts[0], pts[1], pts[2]), p1(pts[3], pts[4], pts[5]), p2(pts[6], pts[7], pts[8]);
  const Vec3 e1 = p1 - p0, e2 = p2 - p0;
  const Vec3 n = e1.cross(e2).norm();
  Vec3 lo = p0, hi = p0;
  for (int i = 0; i < p.n_points; ++i) {
    const Vec3 q(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    lo.x = std::min(lo.x, q.x); lo.y = std::min(lo.y, q.y); lo.z = std::min(lo.z, q.z);
    hi.x = std::max(hi.x, q.x); hi.y = std::max(hi.y, q.y); hi.z = std::max(hi.z, q.z);
  }
  const double tol = 1e-6;
  d.plane_flags = (hi.x - lo.x > tol ? 1 : 0) | (hi.y - lo.y > tol ? 2 : 0) | (hi.z - lo.z > tol ? 4 : 0);
  const double l1 = e1.length(), l2 = e2.length();
  const Vec3 e1n = e1 / l1, e2n = e2 / l2;
  const double v[20] = {p0.x, p0.y, p0.z, n.x, n.y, n.z, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z,
                        e1n.x, e1n.y, e1n.z, l1, e2n.x, e2n.y, e2n.z, l2};
  std::memcpy(d.xf, v, sizeof v);
}

// Decomposition half of Mat4::interpolate (Transform.cpp:239-247) and the t-independent part of
// Quaternion::slerp (Quaternion.h:73-105).
DMotion prepare_motion(const mirt_prim &p, DPrim &d) {
  const Mat4 a = mat_of(p.m0), b = mat_of(p.m1);
  DMotion mo;
  std::memset(&mo, 0, sizeof mo);
  const Vec3 ta = a.extract_translation(), tb = b.extract_translation();
  const Vec3 sa = a.extract_scale(), sb = b.extract_scale();
  const Quaternion q0 = a.extract_rotation().norm();
  Quaternion q1 = b.extract_rotation().norm();
  double dot = q0.dot(q1);
  if (dot < 0.0) {
    q1 = Quaternion(-q1.w, -q1.x, -q1.y, -q1.z);
    dot = -dot;
  }
  const Vec3 dt = tb - ta, ds = sb - sa;
  const double t0[3] = {ta.x, ta.y, ta.z}, dtt[3] = {dt.x, dt.y, dt.z};
  const double s0[3] = {sa.x, sa.y, sa.z}, dss[3] = {ds.x, ds.y, ds.z};
  std::memcpy(mo.t0, t0, sizeof t0);
  std::memcpy(mo.dt, dtt, sizeof dtt);
  std::memcpy(mo.s0, s0, sizeof s0);
  std::memcpy(mo.ds, dss, sizeof dss);
  mo.q0[0] = q0.w; mo.q0[1] = q0.x; mo.q0[2] = q0.y; mo.q0[3] = q0.z;
  mo.lerp = dot > 0.9995 ? 1 : 0;
  if (mo.lerp) {
    mo.dq[0] = q1.w - q0.w; mo.dq[1] = q1.x - q0.x; mo.dq[2] = q1.y - q0.y; mo.dq[3] = q1.z - q0.z;
  } else {
    mo.dq[0] = q1.w; mo.dq[1] = q1.x; mo.dq[2] = q1.y; mo.dq[3] = q1.z;
    dot = std::min(1.0, std::max(-1.0, dot));
    mo.theta0 = std::acos(dot);
    mo.inv_sin0 = 1.0 / std::sin(mo.theta0);
  }
  mo.dot = dot;
  const bool fixed_rs = mo.lerp && mo.dq[0] == 0.0 && mo.dq[1] == 0.0 && mo.dq[2] == 0.0 && mo.dq[3] == 0.0 &&
                        ds.x == 0.0 && ds.y == 0.0 && ds.z == 0.0;
  if (fixed_rs) {
    // rotation and scale do not change over the shutter: only the translation is interpolated
    const Mat4 rs = Mat4::compose(Vec3(0, 0, 0), q0.norm(), sa);
    const Mat4 inv = rs.inverse();
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        mo.m[r * 3 + c] = rs.m[r][c];
        mo.minv[r * 3 + c] = inv.m[r][c];
      }
    d.motion = MOTION_TRANSLATE;
  } else {
    d.motion = MOTION_GENERAL;
  }
  return mo;
}

// The device record of primitive p: kind, flags, the static maps (a moving primitive: the maps at shutter
// time 0.5, built the way the reference builds them per ray: interpolate, then invert, Sphere.cpp:9-11)
// and the shape parameters.  A moving primitive's motion record goes to *mo and the first three rows of
// its matrix_t0 / matrix_t1 to ends[0 .. 23] (its box is the union of the boxes at the two, Sphere.cpp:85-99);
// its motion_index is left for the caller to set.
void prepare_prim(const mirt_scene_desc &s, const mirt_prim &p, DPrim &d, DMotion *mo, double *ends) {
  std::memset(&d, 0, sizeof d);
  d.type = p.type;
  d.visible = p.visible;
  d.material = p.material;
  d.motion = MOTION_NONE;
  std::memcpy(d.param, p.param, sizeof d.param);
  if (p.type == MIRT_PLANE) {  // no transform, no motion (Plane.cpp:8-92)
    prepare_plane(s, p, d);
    return;
  }
  if (p.has_motion) {
    const Transform mid(Mat4::interpolate(mat_of(p.m0), mat_of(p.m1), 0.5));
    store_3x4(d.xf, mid.world_to_object);
    store_3x4(d.xf + 12, mid.object_to_world);
    *mo = prepare_motion(p, d);
    store_3x4(ends, mat_of(p.m0));
    store_3x4(ends + 12, mat_of(p.m1));
  } else {
    store_3x4(d.xf, mat_of(p.w2o));
    store_3x4(d.xf + 12, mat_of(p.o2w));
  }
}

SpawnKinds spawn_kinds(const mirt_scene_desc &s) {
  SpawnKinds k;
  for (int i = 0; i < s.n_materials; ++i) {
    const mirt_material &m = s.materials[i];
    const bool refl = m.reflectivity > 0.0, glass = m.transparency > 0.0, lobe = refl && m.glossiness < 0.94;
    if (refl && !lobe && !glass) k.mirror = true;
    if (lobe && !glass) k.lobe = true;
    if (glass && !refl) k.glass = true;
    if (refl && !lobe && glass) k.mirror_glass = true;
    if (lobe && glass) k.lobe_glass = true;
  }
  return k;
}

int max_children(const SpawnKinds &k, int glossy_samples) {
  const int g = glossy_samples > 1 ? glossy_samples : 1;
  int m = 1;
  if (k.mirror) m = std::max(m, 1);
  if (k.lobe) m = std::max(m, g);
  if (k.glass) m = std::max(m, 2);
  if (k.mirror_glass) m = std::max(m, 3);
  if (k.lobe_glass) m = std::max(m, g + 2);
  return m;
}

constexpr size_t kCountBytes = (size_t)QUEUE_SHARDS * QUEUE_COUNT_STRIDE * sizeof(uint32_t);
// share of the static array set aside for the control block (statistics counters, then 16 KiB of queue
// tails per primary launch and ray-tree level): 4 MiB = 255 launch-levels
constexpr size_t kFixedControlBytes = (size_t)4 << 20;

constexpr size_t kRayBytes = 10 * sizeof(double) + sizeof(uint64_t) + sizeof(uint32_t);  // one queued ray

// Lays the queue's arrays out over its memory; `at` != NULL: over that memory instead (never allocates).
hipError_t ensure_queue(mirt_scene::WorkSet *sc, int which, void *at = nullptr) {
  RayQueueMem &rq = sc->queues[which];
  const size_t cap = sc->ray_shard_cap * QUEUE_SHARDS;
  if (at) rq.mem.borrow(at, cap * kRayBytes);
  else if (rq.q.shard_cap == sc->ray_shard_cap && rq.mem.p) return hipSuccess;
  hipError_t e = rq.mem.reserve(cap * kRayBytes);
  if (e != hipSuccess) return e;
  char *p = rq.mem.as<char>();
  double **fields[10] = {&rq.q.ox, &rq.q.oy, &rq.q.oz, &rq.q.dx, &rq.q.dy, &rq.q.dz, &rq.q.time, &rq.q.wr, &rq.q.wg, &rq.q.wb};
  for (auto f : fields) {
    *f = reinterpret_cast<double *>(p);
    p += cap * sizeof(double);
  }
  rq.q.key = reinterpret_cast<uint64_t *>(p);
  p += cap * sizeof(uint64_t);
  rq.q.pixel = reinterpret_cast<uint32_t *>(p);
  rq.q.shard_cap = (uint32_t)sc->ray_shard_cap;
  return hipSuccess;
}

hipError_t ensure_pinned(mirt_scene::WorkSet *sc, size_t bytes) {
  if (bytes <= sc->pinned_bytes) return hipSuccess;
  if (sc->fixed) return hipErrorOutOfMemory;  // sized once, when the scene was created
  if (sc->pinned) (void)hipHostFree(sc->pinned);
  sc->pinned = nullptr;
  sc->pinned_bytes = 0;
  hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&sc->pinned), bytes, hipHostMallocDefault);
  if (e == hipSuccess) sc->pinned_bytes = bytes;
  return e;
}

size_t env_size(const char *name, size_t dflt) {
  const char *v = std::getenv(name);
  if (!v || !*v) return dflt;
  const long long x = std::atoll(v);
  return x > 0 ? (size_t)x : dflt;
}

bool env_flag(const char *name, bool dflt) {  // "0" switches off, anything else on
  const char *v = std::getenv(name);
  if (!v || !*v) return dflt;
  return !(v[0] == '0' && v[1] == 0);
}

// Camera basis of Camera::get_ray (Camera.cpp:32-58, :91-92).
DCamera device_camera(const mirt_camera &c, int width, int height) {
  DCamera d;
  std::memset(&d, 0, sizeof d);
  const Vec3 gaze(c.gaze[0], c.gaze[1], c.gaze[2]), up(c.up[0], c.up[1], c.up[2]);
  const Vec3 w = -gaze.norm();
  const Vec3 u = up.cross(w).norm();
  const Vec3 v = w.cross(u).norm();
  const double aspect = static_cast<double>(width) / height;
  const double scale_factor = 0.001;
  for (int k = 0; k < 3; ++k) {
    d.location[k] = c.location[k];
    d.gaze[k] = c.gaze[k];
    d.w[k] = w[k];
    d.u[k] = u[k];
    d.v[k] = v[k];
  }
  d.viewport_h = c.sensor_height * scale_factor;
  d.viewport_w = d.viewport_h * aspect;
  d.image_plane_dist = c.focal_length * scale_factor;
  d.focus_distance = c.focus_distance;
  d.dof = (c.dof_enabled && c.aperture_fstop > 0.0) ? 1 : 0;
  d.aperture_radius = d.dof ? (c.focal_length * scale_factor) / (2.0 * c.aperture_fstop) : 0.0;
  return d;
}

hipError_t get_event(mirt_scene::WorkSet *sc, size_t &used, hipEvent_t &e) {
  if (used == sc->events.size()) {
    if (sc->fixed) return hipErrorOutOfMemory;  // its two events were created with the scene
    hipEvent_t n;
    hipError_t err = hipEventCreate(&n);
    if (err != hipSuccess) return err;
    sc->events.push_back(n);
  }
  e = sc->events[used++];
  return hipSuccess;
}

}  // namespace

MIRT_API int mirt_abi_version(void) { return MIRT_ABI_VERSION; }

MIRT_API const char *mirt_last_error(void) { return g_error.c_str(); }

MIRT_API int mirt_device_count(void) {
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(MIRT_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

MIRT_API void mirt_config_defaults(mirt_config *c) {
  if (!c) return;
  std::memset(c, 0, sizeof *c);
  c->max_ray_depth = 12;
  c->shadow_samples = 16;
  c->glossy_samples = 0;
  c->enable_motion_blur = 1;
  c->enable_textures = 1;
  c->use_adaptive_epsilon = 1;
  c->tone_mapping_mode = MIRT_TONEMAP_NONE;
  c->enable_gamma_correction = 1;
  c->light_intensity_factor = 0.2;
  c->ambient_factor = 1.0;
  c->ray_offset_epsilon = 0.001;
  c->adaptive_epsilon_scale = 0.0001;
  c->pure_glass_threshold = 0.99;
  c->exposure = 1.0;
  c->gamma = 2.2;
  c->seed = 0;
  c->sampling = MIRT_SAMPLING_JITTER;
  c->kernel_timing = 0;
}

MIRT_API void mirt_scene_destroy(mirt_scene *scene) {
  if (!scene) return;
  (void)hipSetDevice(scene->device);
  delete scene;
}

MIRT_API int mirt_scene_set_traversal(mirt_scene *scene, int mode) {
  if (!scene) return fail(MIRT_ERR_INVALID, "NULL scene");
  const bool can_order = scene->bvh_depth < WALK_STACK;  // the ordered walk's per-lane stack
  const bool can_wide = 3 * scene->wide_depth + 4 <= WIDE_STACK_LDS + WIDE_STACK_DEEP;  // the wide walk's: a visit pops one entry and pushes at most four
  if (scene->builder == MIRT_BUILDER_DEVICE) {
    // a tree built on the device exists as the wide walk's arrays only
    if (mode != MIRT_TRAVERSAL_AUTO && mode != MIRT_TRAVERSAL_WIDE)
      return fail(MIRT_ERR_UNSUPPORTED, "a scene whose BVH was built on the device has the wide walk only");
    if (!can_wide) return fail(MIRT_ERR_UNSUPPORTED, "BVH deeper than the wide walk's stack");
    scene->walk = WALK_WIDE;
    return MIRT_OK;
  }
  if (mode == MIRT_TRAVERSAL_AUTO) {
    // the packet walk visits the union of its 64 rays' paths: measured faster up to several hundred
    // primitives, slower from about a thousand; above that the wide walk (same closest hit, exact
    // ties may resolve differently) is 2x faster than the per-lane walk in the reference's order
    // on a million primitives.  MIRT_TRAVERSAL=packet|lane|ordered|wide overrides the choice.
    const char *env = std::getenv("MIRT_TRAVERSAL");
    scene->walk = (size_t)scene->n_prims <= env_size("MIRT_PACKET_MAX_PRIMS", 512) ? WALK_PACKET
                  : can_wide                                                        ? WALK_WIDE
                                                                                    : WALK_LANE;
    if (env && std::strcmp(env, "packet") == 0) scene->walk = WALK_PACKET;
    if (env && std::strcmp(env, "lane") == 0) scene->walk = WALK_LANE;
    if (env && std::strcmp(env, "ordered") == 0 && can_order) scene->walk = WALK_ORDERED;
    if (env && std::strcmp(env, "wide") == 0 && can_wide) scene->walk = WALK_WIDE;
  } else if (mode == MIRT_TRAVERSAL_PACKET) {
    scene->walk = WALK_PACKET;
  } else if (mode == MIRT_TRAVERSAL_LANE) {
    scene->walk = WALK_LANE;
  } else if (mode == MIRT_TRAVERSAL_ORDERED) {
    if (!can_order) return fail(MIRT_ERR_UNSUPPORTED, "BVH deeper than the ordered walk's stack (%d)", WALK_STACK);
    scene->walk = WALK_ORDERED;
  } else if (mode == MIRT_TRAVERSAL_WIDE) {
    if (!can_wide) return fail(MIRT_ERR_UNSUPPORTED, "BVH deeper than the wide walk's stack");
    scene->walk = WALK_WIDE;
  } else {
    return fail(MIRT_ERR_INVALID, "unknown traversal mode %d", mode);
  }
  return MIRT_OK;
}

namespace {
int scene_create(const mirt_scene_desc *desc, int device, const mirt_build_options &opts, mirt_scene **out);
}

// No exception leaves the library: the host-side tables of a scene are std::vectors, and a scene of
// 10^6 primitives on a small host can run out of memory while they are filled.
MIRT_API int mirt_scene_create_with(const mirt_scene_desc *desc, int device, const mirt_build_options *options,
                                    mirt_scene **out) {
  if (!out) return fail(MIRT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  mirt_build_options opts;
  std::memset(&opts, 0, sizeof opts);
  if (options) opts = *options;
  if (opts.builder != MIRT_BUILDER_HOST && opts.builder != MIRT_BUILDER_DEVICE)
    return fail(MIRT_ERR_INVALID, "unknown BVH builder %d", opts.builder);
  if (opts.leaf_size < 0 || opts.leaf_size > 4) return fail(MIRT_ERR_INVALID, "leaf_size must be 0 (default) or 1 .. 4");
  try {
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = scene_create(desc, device, opts, out);
    if (rc == MIRT_OK)
      (*out)->create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
  } catch (const std::bad_alloc &) {
    *out = nullptr;
    return fail(MIRT_ERR_MEMORY, "out of host memory while preparing the scene");
  } catch (const std::exception &e) {
    *out = nullptr;
    return fail(MIRT_ERR_INVALID, "scene preparation failed: %s", e.what());
  }
}

MIRT_API int mirt_scene_create(const mirt_scene_desc *desc, int device, mirt_scene **out) {
  return mirt_scene_create_with(desc, device, nullptr, out);
}

namespace {
int scene_create(const mirt_scene_desc *desc, int device, const mirt_build_options &opts, mirt_scene **out) {
  const bool device_tree = opts.builder == MIRT_BUILDER_DEVICE;
  if (int rc = validate(desc, device_tree)) return rc;
  if (device_tree)
    for (int i = 0; i < desc->n_prims; ++i)
      if (desc->prims[i].type == MIRT_PLANE)
        return fail(MIRT_ERR_UNSUPPORTED,
                    "the device BVH builder takes no planes (primitive %d): a flat box is never hit "
                    "(BoundingBox.h:27), so which planes are visible depends on the tree; build this scene's tree "
                    "on the host", i);
  int n_dev = 0;
  HIP_TRY(hipGetDeviceCount(&n_dev));
  if (device < 0 || device >= n_dev) return fail(MIRT_ERR_DEVICE, "device %d not available (%d visible)", device, n_dev);
  HIP_TRY(hipSetDevice(device));

  mirt_scene *sc = new (std::nothrow) mirt_scene;
  if (!sc) return fail(MIRT_ERR_MEMORY, "out of host memory");
  struct Guard {
    mirt_scene *s;
    ~Guard() { delete s; }
  } guard{sc};
  sc->device = device;
  sc->builder = opts.builder;
  HIP_TRY(hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&sc->null_stream_mark, hipEventDisableTiming));
  {
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    sc->resident_grid = (unsigned)env_size("MIRT_RESIDENT_BLOCKS", (size_t)std::max(cus, 1) * 3);
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0)
      sc->wall_clock_khz = khz;
    const char *st = std::getenv("MIRT_STREAM");
    sc->stream_kernels = !(st && st[0] == '0');
  }

  const mirt_scene_desc &s = *desc;
  if (s.n_prims > LEAF_PRIM_MASK) return fail(MIRT_ERR_UNSUPPORTED, "more than %d primitives", LEAF_PRIM_MASK);

  std::vector<DPrim> prims(s.n_prims);
  std::vector<DMotion> &motion = sc->motion_host;
  std::vector<double> &ends = sc->motion_ends_host;
  std::vector<double> transparency(s.n_prims);
  sc->prim_type.resize(s.n_prims);
  sc->prim_material.resize(s.n_prims);
  sc->prim_motion_index.assign(s.n_prims, -1);
  bool any_maps = false, any_hit = true;
  {
    // Device records of the primitives.  Large scenes: contiguous ranges on separate threads, each collecting
    // the motion records of its range; the ranges' records are then appended in order, so the arrays are the ones
    // the sequential loop would write.
    struct Range {
      int a = 0, b = 0;
      std::vector<DMotion> motion;
      std::vector<double> ends;
      bool any_hit = true;
    };
    const int n_threads = s.n_prims >= 65536 ? (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u) : 1;
    std::vector<Range> ranges((size_t)n_threads);
    auto work = [&](Range &r) {
      for (int i = r.a; i < r.b; ++i) {
        const mirt_prim &p = s.prims[i];
        DPrim &d = prims[i];
        DMotion mo;
        double e[24];
        prepare_prim(s, p, d, &mo, e);
        if (d.motion != MOTION_NONE) {
          d.motion_index = (int32_t)r.motion.size();  // within the range; offset added below
          r.motion.push_back(mo);
          r.ends.insert(r.ends.end(), e, e + 24);
        }
        sc->prim_type[i] = (uint8_t)p.type;
        sc->prim_material[i] = p.material;
        const double tr = s.materials[p.material].transparency;
        transparency[i] = tr;
        if (tr > 0.0 && tr != 1.0) r.any_hit = false;  // attenuates: shadow segments must be chained through it
      }
    };
    for (int t = 0; t < n_threads; ++t) {
      ranges[t].a = (int)((long long)s.n_prims * t / n_threads);
      ranges[t].b = (int)((long long)s.n_prims * (t + 1) / n_threads);
    }
    if (n_threads == 1) {
      work(ranges[0]);
    } else {
      std::vector<std::thread> pool;
      for (int t = 0; t < n_threads; ++t) pool.emplace_back(work, std::ref(ranges[t]));
      for (auto &th : pool) th.join();
    }
    for (const Range &r : ranges) {
      const int32_t base = (int32_t)motion.size();
      for (int i = r.a; i < r.b && !r.motion.empty(); ++i)
        if (prims[i].motion != MOTION_NONE) {
          prims[i].motion_index += base;
          sc->prim_motion_index[i] = prims[i].motion_index;
        }
      motion.insert(motion.end(), r.motion.begin(), r.motion.end());
      ends.insert(ends.end(), r.ends.begin(), r.ends.end());
      any_hit = any_hit && r.any_hit;
    }
  }
  std::vector<mirt_material> materials(s.materials, s.materials + s.n_materials);
  for (const auto &m : materials)
    if ((m.has_texture && m.texture >= 0) || m.normal_map >= 0 || m.bump_map >= 0) any_maps = true;
  sc->kinds = spawn_kinds(s);
  sc->n_prims = s.n_prims;

  std::vector<DTexture> textures(s.n_textures);
  sc->texels.resize(s.n_textures);
  for (int i = 0; i < s.n_textures; ++i) {
    const mirt_texture &t = s.textures[i];
    const size_t bytes = (size_t)t.width * t.height * 3 * sizeof(uint16_t);
    HIP_TRY(sc->texels[i].reserve(bytes));
    HIP_TRY(hipMemcpy(sc->texels[i].p, t.rgb, bytes, hipMemcpyHostToDevice));
    textures[i].rgb = sc->texels[i].as<uint16_t>();
    textures[i].width = t.width;
    textures[i].height = t.height;
    textures[i].max_value = t.max_value;
    textures[i].inv_max = 1.0 / t.max_value;
    textures[i].pad = 0;
  }
  sc->host_lights.assign(s.lights, s.lights + s.n_lights);
  sc->point_lights = true;
  for (const mirt_light &L : sc->host_lights)
    if (L.type == MIRT_LIGHT_AREA) sc->point_lights = false;

  HIP_TRY(upload(sc->prims, prims));
  HIP_TRY(upload(sc->motion, motion));
  HIP_TRY(upload(sc->motion_ends, ends));
  HIP_TRY(upload(sc->materials, materials));
  HIP_TRY(upload(sc->transparency, transparency));
  HIP_TRY(upload(sc->lights, sc->host_lights));
  HIP_TRY(upload(sc->textures, textures));
  HIP_TRY(sc->light_frame.reserve(std::max<size_t>(sizeof(DLightFrame) * s.n_lights, 16)));
  sc->light_frame_host.resize(s.n_lights);

  // ---- the tree ------------------------------------------------------------------------------------
  const auto tree_t0 = std::chrono::steady_clock::now();
  DScene &d = sc->d;
  if (device_tree) {
    // the reference-order walks have no tree here: one empty leaf stands in for their arrays
    std::vector<DNode> nodes(1);
    std::memset(nodes.data(), 0, sizeof(DNode));
    nodes[0].skip = 1;
    HIP_TRY(upload(sc->nodes, nodes));
    HIP_TRY(upload(sc->leaf_prims, std::vector<int32_t>()));
    HIP_TRY(upload(sc->leaf_tagged, std::vector<int32_t>()));
    sc->bvh_depth = 0;
    sc->dbvh.reset(new DeviceBvh);
    const uint32_t leaf = opts.leaf_size ? (uint32_t)opts.leaf_size : (uint32_t)env_size("MIRT_DEVICE_LEAF", 2);
    const int be = sc->dbvh->build(sc->stream, sc->prims.as<DPrim>(), sc->motion_ends.as<double>(),
                                   sc->transparency.as<double>(), (uint32_t)s.n_prims, leaf);
    if (be != 0)
      return fail(be == (int)hipErrorOutOfMemory ? MIRT_ERR_MEMORY : MIRT_ERR_DEVICE, "device BVH build: %s",
                  sc->dbvh->error.c_str());
    sc->wide_depth = sc->dbvh->wide_depth;
    d.wide = sc->dbvh->wide;
    d.leaves = sc->dbvh->leaves;
    d.n_nodes = 1;
    if (int rc = mirt_scene_set_traversal(sc, MIRT_TRAVERSAL_AUTO)) return rc;
  } else {
    std::vector<DNode> nodes(s.n_nodes);
    int bvh_depth = 0;
    if (fill_skip(s, nodes, 0, 0, bvh_depth) != s.n_nodes)
      return fail(MIRT_ERR_INVALID, "BVH nodes are not one depth-first tree rooted at node 0");
    sc->bvh_depth = bvh_depth;
    std::vector<int32_t> leaf(s.leaf_prims, s.leaf_prims + s.n_leaf_prims);
    std::vector<int32_t> tagged(leaf);
    for (auto &t : tagged)
      t |= (s.prims[t].type << LEAF_KIND_SHIFT) | (transparency[t] == 1.0 ? LEAF_CLEAR_BIT : 0);
    // a leaf's first primitive rides in its node record: the packet walk then reaches it without a trip to leaf_tagged
    for (auto &n : nodes)
      if (n.first >= 0) n.order = n.count > 0 ? tagged[n.first] : 0;
    HIP_TRY(upload(sc->nodes, nodes));
    HIP_TRY(upload(sc->leaf_prims, leaf));
    HIP_TRY(upload(sc->leaf_tagged, tagged));
    WideBuilder wb{s, nodes, tagged, {}, {}};
    wb.wide.reserve(s.n_nodes / 3 + 1);
    wb.leaves.reserve(s.n_nodes / 2 + 1);
    wb.build(0, 0);
    sc->wide_depth = wb.depth;
    mirt_scene_set_traversal(sc, MIRT_TRAVERSAL_AUTO);  // needs n_prims, bvh_depth, wide_depth
    HIP_TRY(upload(sc->wide, wb.wide));
    HIP_TRY(upload(sc->leaves, wb.leaves));
    d.wide = sc->wide.as<DWide>();
    d.leaves = sc->leaves.as<DLeaf>();
    d.n_nodes = s.n_nodes;
    sc->n_wide = (uint32_t)wb.wide.size();
    sc->n_leaves = (uint32_t)wb.leaves.size();
  }
  sc->tree_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tree_t0).count();

  // Everything mirt_render_tiles_noalloc will need, so that the call itself allocates nothing: the
  // address of this GPU's static array, the read-back buffer of the control block and the two
  // events that bracket a frame.
  {
    void *sp = nullptr;
    const int se = scratch_address(device, &sp);
    if (se != 0)
      return fail(MIRT_ERR_DEVICE, "hipGetSymbolAddress(scratch) on device %d: %s", device,
                  hipGetErrorString(static_cast<hipError_t>(se)));
    sc->scratch = static_cast<char *>(sp);
    mirt_scene::WorkSet &fw = sc->fixed_work;
    HIP_TRY(ensure_pinned(&fw, kFixedControlBytes));
    for (int k = 0; k < 2; ++k) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreate(&ev));
      fw.events.push_back(ev);
    }
    fw.fixed = true;
  }

  d.nodes = sc->nodes.as<DNode>();
  d.leaf_prims = sc->leaf_prims.as<int32_t>();
  d.leaf_tagged = sc->leaf_tagged.as<int32_t>();
  d.prims = sc->prims.as<DPrim>();
  d.motion = sc->motion.as<DMotion>();
  d.materials = sc->materials.as<mirt_material>();
  d.prim_transparency = sc->transparency.as<double>();
  d.lights = sc->lights.as<mirt_light>();
  d.textures = sc->textures.as<DTexture>();
  d.n_prims = s.n_prims;
  d.n_lights = s.n_lights;
  d.n_textures = s.n_textures;
  for (int k = 0; k < 3; ++k) d.background[k] = s.background_color[k] * s.background_strength;
  d.kinds = 0;
  for (int i = 0; i < s.n_prims; ++i) d.kinds |= 1 << s.prims[i].type;
  d.any_motion = motion.empty() ? 0 : 1;
  d.any_maps = any_maps ? 1 : 0;
  d.max_children = 0;
  {
    const char *ah = std::getenv("MIRT_SHADOW_ANY_HIT");  // "0": chain closest hits even where any-hit is exact
    d.shadow_any_hit = (any_hit && !(ah && ah[0] == '0')) ? 1 : 0;
  }
  {
    // Two surfaces at (nearly) one distance -- a cylinder standing on a plane -- go to whichever the reference
    // tests later and accepts (Bvh.cpp:312-332 with each intersector's own comparison): the wide walk flags rays
    // whose two best hits lie within this relative band and walks them again over the reference's tree in the
    // reference's order.  Needs that tree; MIRT_TIE_BAND=0 switches it off.
    const char *tb = std::getenv("MIRT_TIE_BAND");
    const double band = !device_tree ? (tb && *tb ? std::atof(tb) : 1e-9) : 0.0;
    d.tie_hi = 1.0 + band;
    d.tie_lo = (1.0 - band) / (1.0 + band);
  }

  guard.s = nullptr;
  *out = sc;
  return MIRT_OK;
}
}  // namespace

MIRT_API int mirt_scene_get_info(const mirt_scene *scene, mirt_scene_info *info) {
  if (!scene || !info) return fail(MIRT_ERR_INVALID, "NULL argument");
  std::memset(info, 0, sizeof *info);
  info->builder = scene->builder;
  info->traversal = scene->walk == WALK_PACKET ? MIRT_TRAVERSAL_PACKET
                    : scene->walk == WALK_LANE ? MIRT_TRAVERSAL_LANE
                    : scene->walk == WALK_ORDERED ? MIRT_TRAVERSAL_ORDERED
                                                  : MIRT_TRAVERSAL_WIDE;
  info->n_prims = scene->n_prims;
  info->n_wide_nodes = (int32_t)(scene->dbvh ? scene->dbvh->n_wide : scene->n_wide);
  info->n_leaves = (int32_t)(scene->dbvh ? scene->dbvh->n_leaves : scene->n_leaves);
  info->wide_depth = scene->wide_depth;
  info->bvh_depth = scene->bvh_depth;
  info->shadow_any_hit = scene->d.shadow_any_hit;
  info->create_ms = scene->create_ms;
  info->tree_ms = scene->tree_ms;
  info->refit_ms = scene->dbvh ? scene->dbvh->refit_ms : 0.0;
  return MIRT_OK;
}

// New records for primitives [first, first + count) of a scene whose tree was built on the device.
MIRT_API int mirt_scene_update_prims(mirt_scene *scene, int first, int count, const mirt_prim *prims) {
  if (!scene || (!prims && count > 0)) return fail(MIRT_ERR_INVALID, "NULL argument");
  if (scene->builder != MIRT_BUILDER_DEVICE)
    return fail(MIRT_ERR_UNSUPPORTED, "only a scene whose BVH was built on the device can be refitted; flatten the "
                "changed scene and create it again");
  if (first < 0 || count < 0 || (long long)first + count > scene->n_prims)
    return fail(MIRT_ERR_INVALID, "primitives [%d, %d) outside [0, %d)", first, first + count, scene->n_prims);
  for (const auto &w : scene->work)
    if (w.pending) return fail(MIRT_ERR_INVALID, "a frame is in flight: call mirt_render_tiles_end first");
  if (count == 0) return MIRT_OK;
  HIP_TRY(hipSetDevice(scene->device));
  try {
    std::vector<DPrim> recs((size_t)count);
    bool moved = false;
    mirt_scene_desc none;
    std::memset(&none, 0, sizeof none);
    for (int k = 0; k < count; ++k) {
      const mirt_prim &p = prims[k];
      const int i = first + k;
      const int mi = scene->prim_motion_index[i];
      if (p.type != scene->prim_type[i] || p.material != scene->prim_material[i] || (p.has_motion != 0) != (mi >= 0))
        return fail(MIRT_ERR_INVALID, "primitive %d: kind, material and has_motion cannot change in an update", i);
      DMotion mo;
      double e[24];
      prepare_prim(none, p, recs[k], &mo, e);
      if (mi >= 0) {
        recs[k].motion_index = mi;
        scene->motion_host[mi] = mo;
        std::memcpy(&scene->motion_ends_host[24 * (size_t)mi], e, sizeof e);
        moved = true;
      }
    }
    HIP_TRY(hipMemcpy(scene->prims.as<DPrim>() + first, recs.data(), recs.size() * sizeof(DPrim), hipMemcpyHostToDevice));
    if (moved) {
      HIP_TRY(upload(scene->motion, scene->motion_host));
      HIP_TRY(upload(scene->motion_ends, scene->motion_ends_host));
    }
  } catch (const std::bad_alloc &) {
    return fail(MIRT_ERR_MEMORY, "out of host memory");
  }
  return MIRT_OK;
}

MIRT_API int mirt_scene_refit(mirt_scene *scene) {
  if (!scene) return fail(MIRT_ERR_INVALID, "NULL scene");
  if (scene->builder != MIRT_BUILDER_DEVICE || !scene->dbvh)
    return fail(MIRT_ERR_UNSUPPORTED, "only a scene whose BVH was built on the device can be refitted");
  for (const auto &w : scene->work)
    if (w.pending) return fail(MIRT_ERR_INVALID, "a frame is in flight: call mirt_render_tiles_end first");
  HIP_TRY(hipSetDevice(scene->device));
  const int e = scene->dbvh->refit(scene->stream, scene->motion_ends.as<double>());
  if (e != 0) return fail(MIRT_ERR_DEVICE, "device BVH refit: %s", scene->dbvh->error.c_str());
  return MIRT_OK;
}

MIRT_API int64_t mirt_tile_buffer_pixels(int width, int height, int tile_w, int tile_h, int rank, int world) {
  if (width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0 || world <= 0 || rank < 0 || rank >= world) return -1;
  const int64_t tiles = (int64_t)((width + tile_w - 1) / tile_w) * ((height + tile_h - 1) / tile_h);
  const int64_t mine = tiles > rank ? (tiles - rank + world - 1) / world : 0;
  return mine * tile_w * tile_h;
}

MIRT_API int mirt_tile_owner(int tx, int ty, int tiles_x, int world) {
  if (tx < 0 || ty < 0 || tiles_x <= 0 || tx >= tiles_x || world <= 0) return -1;
  return tile_owner(tx, ty, tiles_x, world, tile_stagger(tiles_x, world));
}

namespace {

// The orthonormal basis around an area light's normal (Light.h:44-52), same expressions in the same
// order as the reference evaluates per sample.
void device_light_basis(const mirt_light &L, DLightFrame &lf) {
  auto unit = [](Vec3 a) {
    const double len = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return Vec3(a.x / len, a.y / len, a.z / len);
  };
  auto cross = [](Vec3 a, Vec3 b) { return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); };
  Vec3 n(L.normal[0], L.normal[1], L.normal[2]);
  if (n.x * n.x + n.y * n.y + n.z * n.z < 0.1) n = Vec3(0, 0, -1);
  n = unit(n);
  Vec3 right = std::fabs(n.x) > 0.9 ? Vec3(0, 1, 0) : Vec3(1, 0, 0);
  right = unit(cross(n, right));
  const Vec3 up = unit(cross(n, right));
  lf.right[0] = right.x; lf.right[1] = right.y; lf.right[2] = right.z;
  lf.up[0] = up.x; lf.up[1] = up.y; lf.up[2] = up.z;
}

// Geometry of one rank's share of a frame.
struct FrameShape {
  DFrame fr;
  size_t n_pixels;
  uint64_t total_blocks;  // 16x16 pixel blocks of this rank's tiles
  int64_t tiles;
};

FrameShape frame_shape(const mirt_camera &camera, const mirt_config &config, int width, int height, int spp, int tile_w,
                       int tile_h, int rank, int world) {
  FrameShape f;
  DFrame &fr = f.fr;
  std::memset(&fr, 0, sizeof fr);
  fr.cam = device_camera(camera, width, height);
  fr.cfg = config;
  fr.width = width;
  fr.height = height;
  fr.spp = spp;
  fr.tile_w = tile_w;
  fr.tile_h = tile_h;
  fr.tiles_x = (width + tile_w - 1) / tile_w;
  fr.tiles_y = (height + tile_h - 1) / tile_h;
  fr.rank = rank;
  fr.world = world;
  fr.stagger = tile_stagger(fr.tiles_x, world);
  fr.blocks_x = (tile_w + 15) / 16;
  fr.blocks_y = (tile_h + 15) / 16;
  f.tiles = (int64_t)fr.tiles_x * fr.tiles_y;
  fr.n_slots = (int32_t)(f.tiles > rank ? (f.tiles - rank + world - 1) / world : 0);
  f.n_pixels = (size_t)fr.n_slots * tile_w * tile_h;
  fr.plane = (uint32_t)f.n_pixels;
  f.total_blocks = (uint64_t)fr.n_slots * fr.blocks_x * fr.blocks_y;
  return f;
}

// Enqueues the frame recorded in `ws` on its stream: fills, every launch of every depth, resolve and
// the read-back of the control block.  Nothing here waits for the device.
int enqueue_frame(mirt_scene *sc, mirt_scene::WorkSet *ws) {
  const mirt_config *config = &ws->config;
  const int spp = ws->spp;
  hipStream_t stream = ws->stream;
  FrameShape shape = frame_shape(ws->camera, ws->config, ws->width, ws->height, spp, ws->tile_w, ws->tile_h, ws->rank,
                                 ws->world);
  DFrame &fr = shape.fr;
  const size_t n_pixels = shape.n_pixels;
  const uint64_t total_blocks = shape.total_blocks;

  // shadow strata per light for this frame (Raytracer.cpp:146-158)
  const int n_lights = sc->d.n_lights;
  if (n_lights && (sc->light_frame_for != config->shadow_samples || !sc->light_frame_valid)) {
    std::vector<DLightFrame> &lf = sc->light_frame_host;  // n_lights entries since the scene was created
    uint32_t dim = MIRT_DIM_SHADOW;
    sc->any_strata = false;
    for (int l = 0; l < n_lights; ++l) {
      const mirt_light &L = sc->host_lights[l];
      int samples = (L.type == MIRT_LIGHT_AREA) ? std::max(1, L.samples) : 1;
      if (config->shadow_samples > 0 && L.type == MIRT_LIGHT_AREA) samples = config->shadow_samples;
      int side = static_cast<int>(std::sqrt(samples));
      if (side * side == 0) side = 1;
      if (side > 1) sc->any_strata = true;
      lf[l].dim = dim;
      lf[l].side = side;
      lf[l].inv_n = 1.0 / (side * side);
      device_light_basis(L, lf[l]);
      dim += 2u * (uint32_t)(side * side);
    }
    // blocking copy from pageable memory: done once per shadow_samples value, not per frame; frames
    // in flight may still read the old table
    for (auto &w : sc->work)
      if (w.pending) HIP_TRY(hipStreamSynchronize(w.stream));
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy(sc->light_frame.p, lf.data(), sizeof(DLightFrame) * n_lights, hipMemcpyHostToDevice));
    sc->light_frame_for = config->shadow_samples;
    sc->light_frame_valid = true;
  }
  fr.light_frame = sc->light_frame.as<DLightFrame>();

  // ---- working memory ------------------------------------------------------------------------
  // A launch of B virtual blocks puts at most ceil(B / QUEUE_SHARDS) of them on one shard, each
  // appending at most 256 * S shadow items and 256 * K rays.  The shadow queue is sized so that a
  // primary launch can never overflow it.  The ray queues and the deeper levels are sized for what
  // scenes do in practice (at most one spawned ray per sample and level on average); a frame that
  // overflows a shard anyway is detected from the tail counters when it is over and rendered again
  // with finer primary launches.
  const int levels = config->max_ray_depth;
  const int K = max_children(sc->kinds, config->glossy_samples);
  const int S = std::max(1, n_lights);
  const uint64_t grid_total = total_blocks * (uint64_t)spp;  // primary grid blocks of the whole frame
  const size_t shard_blocks = (size_t)((grid_total + QUEUE_SHARDS - 1) / QUEUE_SHARDS);
  const size_t min_blocks = std::max<size_t>(1, (size_t)(spp + QUEUE_SHARDS - 1) / QUEUE_SHARDS);  // one pixel block, all samples
  const size_t accum_bytes = std::max<size_t>(n_pixels * 3 * sizeof(double), 64);
  // Large scenes (wide walk): closest hits by a resident grid whose lanes take rays off a work counter,
  // shading by a second launch; the hits of one launch wait in between.  Not with the static working
  // memory, which has no room for them.
  const bool streaming = sc->walk == WALK_WIDE && sc->stream_kernels && !ws->fixed;
  // ... and their shadow items are queued by light (2 to 8 lights): light li owns `shadow_per` of the shards
  uint32_t shadow_groups = 1, shadow_per = QUEUE_SHARDS;
  if (streaming && n_lights >= 2 && n_lights <= 8 && env_flag("MIRT_SHADOW_BY_LIGHT", true)) {
    shadow_groups = (uint32_t)n_lights;
    shadow_per = QUEUE_SHARDS / shadow_groups;
  }
  // most blocks of one launch that append to one shadow shard, and most items one block appends to it
  const size_t shadow_blocks = shadow_groups > 1 ? (size_t)((grid_total + shadow_per - 1) / shadow_per) : shard_blocks;
  const size_t shadow_min_blocks = shadow_groups > 1 ? std::max<size_t>(1, ((size_t)spp + shadow_per - 1) / shadow_per) : min_blocks;
  const size_t block_items = shadow_groups > 1 ? 256 : 256 * (size_t)S;
  if (!ws->fixed) {
    size_t want_r = env_size("MIRT_RAY_QUEUE", (size_t)32 << 20) / QUEUE_SHARDS;
    size_t want_s = env_size("MIRT_SHADOW_QUEUE", (size_t)128 << 20) / QUEUE_SHARDS;
    want_r = std::max(std::min(want_r, shard_blocks * 256 * K), min_blocks * 256 * K);
    want_s = std::max(std::min(want_s, shadow_blocks * block_items * std::max(K, 1)), shadow_min_blocks * block_items);
    if (want_r * QUEUE_SHARDS > 0xfffffff0u || want_s * QUEUE_SHARDS > 0xfffffff0u)
      return fail(MIRT_ERR_UNSUPPORTED, "queue too large");
    if (want_r > ws->ray_shard_cap) ws->ray_shard_cap = want_r;
    if (want_s > ws->shadow_shard_cap) ws->shadow_shard_cap = want_s;
    HIP_TRY(ws->shadow_items.reserve(ws->shadow_shard_cap * QUEUE_SHARDS * sizeof(DShadowItem)));
    const void *before = ws->accum.p;
    HIP_TRY(ws->accum.reserve(accum_bytes));
    if (ws->accum.p != before) ws->accum_clean = false;
    if (levels > 1) {
      HIP_TRY(ensure_queue(ws, 0));
      HIP_TRY(ensure_queue(ws, 1));
    }
  } else {
    // The static array, cut up for this frame: control block | accumulator | shadow queue | two ray
    // queues.  The queues share what the first two leave, in the proportion that lets a primary launch
    // fill them evenly (S shadow items and one ray per sample); a frame whose smallest possible
    // launch -- one pixel block, all samples -- does not fit is refused.
    const size_t acc = (accum_bytes + 255) & ~(size_t)255;
    if (kFixedControlBytes + acc >= MIRT_SCRATCH_BYTES)
      return fail(MIRT_ERR_UNSUPPORTED, "frame of %zu pixels needs %zu bytes of accumulator: more than the %u-byte static "
                  "working memory holds", n_pixels, acc, (unsigned)MIRT_SCRATCH_BYTES);
    const size_t rest = MIRT_SCRATCH_BYTES - kFixedControlBytes - acc;
    const size_t per_entry = (levels > 1 ? 2 * kRayBytes : 0) + (size_t)S * sizeof(DShadowItem);
    size_t cap = rest / (QUEUE_SHARDS * per_entry);
    cap -= cap % 256;
    if (cap < min_blocks * 256 * (size_t)K)
      return fail(MIRT_ERR_UNSUPPORTED, "frame too large for the static working memory: %zu pixels, %d samples, up to %d "
                  "rays per hit leave room for %zu queue entries per shard, %zu needed", n_pixels, spp, K, cap,
                  min_blocks * 256 * (size_t)K);
    ws->ray_shard_cap = cap;
    ws->shadow_shard_cap = cap * (size_t)S;
    char *at = sc->scratch;
    ws->control.borrow(at, kFixedControlBytes);
    at += kFixedControlBytes;
    ws->accum.borrow(at, acc);
    at += acc;
    const size_t shadow_bytes = ws->shadow_shard_cap * QUEUE_SHARDS * sizeof(DShadowItem);
    ws->shadow_items.borrow(at, shadow_bytes);
    at += shadow_bytes;
    if (levels > 1) {
      HIP_TRY(ensure_queue(ws, 0, at));
      at += ws->ray_shard_cap * QUEUE_SHARDS * kRayBytes;
      HIP_TRY(ensure_queue(ws, 1, at));
    }
    ws->accum_clean = false;  // the array keeps whatever the last call on this GPU left in it
  }
  // grid of the launches that consume a queue: fixed, the blocks stride over the queue
  const unsigned consumer_grid = (unsigned)std::min<size_t>(
      env_size("MIRT_CONSUMER_BLOCKS", 4096),
      (size_t)QUEUE_SHARDS * ((std::max(ws->ray_shard_cap, ws->shadow_shard_cap) + 255) / 256));

  const uint64_t shape_key = grid_total * 1000003ull + (uint64_t)levels * 1009ull + (uint64_t)config->glossy_samples;
  // A frame shape seen for the first time is probed: after its first primary launch (with all its levels) has
  // been enqueued, the host waits once and looks at that launch's tails; an overflow restarts the frame with the
  // split that fits before the other launches have been spent on a frame that would have to be redone.
  bool probe = false;
  if (sc->split_for != shape_key) {
    sc->split_for = shape_key;
    sc->split = 1;
    probe = !ws->fixed;
  }
  const bool uv = sc->d.any_maps && config->enable_textures;
  const bool motion = sc->d.any_motion && config->enable_motion_blur && config->sampling == MIRT_SAMPLING_JITTER;

  // pixel blocks per primary launch
  const uint64_t fit = std::min<uint64_t>((uint64_t)(shadow_groups > 1 ? shadow_per : QUEUE_SHARDS) * (ws->shadow_shard_cap / block_items),
                                         (uint64_t)QUEUE_SHARDS * (ws->ray_shard_cap / (size_t)256));
  const uint64_t per_launch = std::max<uint64_t>(1, fit / (uint64_t)spp / sc->split);
  // the launches interleave runs of `run` consecutive blocks (TraceArgs::first): short enough that every launch
  // samples the whole image, long enough that blocks rendered at the same time are neighbours
  const uint64_t run = std::min<uint64_t>(32, per_launch);
  const uint64_t runs_total = (total_blocks + run - 1) / run, runs_per_launch = per_launch / run;
  const size_t chunks = total_blocks ? (size_t)((runs_total + runs_per_launch - 1) / runs_per_launch) : 0;
  if (streaming) {
    const size_t most = std::max<size_t>((size_t)std::min<uint64_t>(per_launch, runs_total * run) * (size_t)spp * 256,
                                         levels > 1 ? ws->ray_shard_cap * QUEUE_SHARDS : 0);
    HIP_TRY(ws->hits.reserve(std::max<size_t>(most, 1) * sizeof(DHit)));
    if (uv) HIP_TRY(ws->hits_uv.reserve(std::max<size_t>(most, 1) * sizeof(DHitUV)));
  }
  // tail counters: per primary launch and level, the rays and the shadow items that level queued
  const size_t stats_bytes = COUNTER_SHARDS * sizeof(DCounters);
  const size_t control_bytes = stats_bytes + chunks * (size_t)std::max(levels, 0) * 2 * kCountBytes;
  if (ws->fixed && control_bytes > kFixedControlBytes)
    return fail(MIRT_ERR_UNSUPPORTED, "frame too large for the static working memory: %zu primary launches x %d levels "
                "need a %zu-byte control block, %zu available", chunks, levels, control_bytes, kFixedControlBytes);
  HIP_TRY(ws->control.reserve(control_bytes));
  HIP_TRY(ensure_pinned(ws, control_bytes));
  DCounters *counters = ws->control.as<DCounters>();
  auto tails = [&](size_t chunk, int level, int what) {  // what: 0 rays, 1 shadow items
    return reinterpret_cast<uint32_t *>(ws->control.as<char>() + stats_bytes +
                                        ((chunk * (size_t)levels + (size_t)level) * 2 + (size_t)what) * kCountBytes);
  };
  const bool kernel_timing = config->kernel_timing != 0 && !ws->fixed;  // per-launch events are created on demand

  size_t ev_used = 0;
  ws->timed.clear();
  ws->launches = 0;
  ws->attempts++;
  ws->chunks = chunks;
  ws->control_bytes = control_bytes;
  ws->per_launch = per_launch;
  HIP_TRY(get_event(ws, ev_used, ws->ev_begin));
  HIP_TRY(get_event(ws, ev_used, ws->ev_end));
  HIP_TRY(hipEventRecord(ws->ev_begin, stream));
  HIP_TRY(hipMemsetAsync(ws->control.p, 0, control_bytes, stream));
  if (n_pixels && !(ws->accum_clean && !ws->fixed && ws->accum_clean_bytes >= n_pixels * 3 * sizeof(double)))
    HIP_TRY(hipMemsetAsync(ws->accum.p, 0, n_pixels * 3 * sizeof(double), stream));
  ws->accum_clean = false;

  unsigned long long *progress = nullptr;
  if (sc->record_progress && !ws->fixed) {  // (the set that never allocates leaves the feed out)
    sc->progress_bands = (ws->height + 15) / 16;
    const size_t bytes = (size_t)(sc->progress_bands + 1) * sizeof(unsigned long long);
    HIP_TRY(sc->progress.reserve(bytes));
    HIP_TRY(hipMemsetAsync(sc->progress.p, 0, bytes, stream));
    HIP_TRY(hipMemsetAsync(sc->progress.p, 0xff, sizeof(unsigned long long), stream));  // earliest start: none yet
    progress = sc->progress.as<unsigned long long>();
  }
  for (size_t c = 0; c < chunks && levels > 0; ++c) {
    // launch c takes the runs c, c + chunks, c + 2 chunks, ...
    const uint32_t nb = (uint32_t)(((runs_total - c + chunks - 1) / chunks) * run);
    for (int depth = 0; depth < levels; ++depth) {
      const bool spawn = depth + 1 < levels;
      TraceArgs a;
      std::memset(&a, 0, sizeof a);
      a.sc = sc->d;
      a.fr = fr;
      if (depth > 0) {
        a.in = ws->queues[depth & 1].q;
        a.in_count = tails(c, depth - 1, 0);
      }
      if (spawn) {
        a.out = ws->queues[(depth + 1) & 1].q;
        a.out_count = tails(c, depth, 0);
      }
      a.shadow = ws->shadow_items.as<DShadowItem>();
      a.shadow_count = tails(c, depth, 1);
      a.shadow_shard_cap = (uint32_t)ws->shadow_shard_cap;
      a.shadow_groups = shadow_groups;
      a.shadow_per = shadow_per;
      a.accum = ws->accum.as<double>();
      a.counters = counters;
      a.first = (uint32_t)c;
      a.n = nb;
      a.stride = (uint32_t)chunks;
      a.group = (uint32_t)run;
      a.total = (uint32_t)total_blocks;
      a.depth = depth;
      a.progress = depth == 0 ? progress : nullptr;
      Timed t{nullptr, nullptr, 0};
      if (kernel_timing) {
        HIP_TRY(get_event(ws, ev_used, t.a));
        HIP_TRY(get_event(ws, ev_used, t.b));
        HIP_TRY(hipEventRecord(t.a, stream));
      }
      const unsigned grid = depth == 0 ? nb * (unsigned)spp : consumer_grid;
      if (streaming) {
        // a spare word of a tail block that this launch owns and nobody reads: the block of the rays it
        // consumes (depth > 0), of the rays it appends (depth 0; appends touch words s * 64 only)
        a.grab = depth > 0 ? tails(c, depth - 1, 0) + 2 : tails(c, 0, 0) + 3;
        a.hits = ws->hits.as<DHit>();
        a.hits_uv = ws->hits_uv.as<DHitUV>();
        launch_closest_stream(depth == 0, uv, motion, sc->resident_grid, stream, a);
        launch_shade(depth == 0, uv, grid, stream, a);
        ws->launches++;
      } else {
        launch_trace(sc->walk, depth == 0, uv, motion, grid, stream, a);
      }
      {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess)
          return fail(MIRT_ERR_DEVICE, "trace launch (depth %d, grid %u): %s", depth, grid, hipGetErrorString(le));
      }
      if (kernel_timing) {
        HIP_TRY(hipEventRecord(t.b, stream));
        ws->timed.push_back(t);
      }
      ws->launches++;

      ShadowArgs sa;
      std::memset(&sa, 0, sizeof sa);
      sa.sc = sc->d;
      sa.fr = fr;
      sa.items = ws->shadow_items.as<DShadowItem>();
      sa.count = a.shadow_count;
      sa.shard_cap = (uint32_t)ws->shadow_shard_cap;
      sa.accum = ws->accum.as<double>();
      sa.counters = counters;
      Timed ts{nullptr, nullptr, 1};
      if (kernel_timing) {
        HIP_TRY(get_event(ws, ev_used, ts.a));
        HIP_TRY(get_event(ws, ev_used, ts.b));
        HIP_TRY(hipEventRecord(ts.a, stream));
      }
      if (streaming) {
        sa.grab = a.shadow_count + 1;  // a spare word of the launch's tail block, zeroed with it
        launch_shadow_stream(sc->d.shadow_any_hit != 0, sc->resident_grid, stream, sa);
      } else {
        launch_shadow(sc->walk, sc->any_strata, sc->d.shadow_any_hit != 0, sc->point_lights, consumer_grid, stream, sa);
      }
      HIP_TRY(hipGetLastError());
      if (kernel_timing) {
        HIP_TRY(hipEventRecord(ts.b, stream));
        ws->timed.push_back(ts);
      }
      ws->launches++;
    }
    if (probe && c == 0 && chunks > 1) {
      const size_t first = (size_t)levels * 2 * kCountBytes;
      HIP_TRY(hipMemcpyAsync(ws->pinned + stats_bytes, ws->control.as<char>() + stats_bytes, first, hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      double worst = 0.0;
      const uint32_t *words = reinterpret_cast<const uint32_t *>(ws->pinned + stats_bytes);
      for (int level = 0; level < levels; ++level)
        for (int sh = 0; sh < QUEUE_SHARDS; ++sh) {
          const size_t at = ((size_t)level * 2) * (kCountBytes / sizeof(uint32_t)) + (size_t)sh * QUEUE_COUNT_STRIDE;
          worst = std::max(worst, std::max((double)words[at] / (double)std::max<size_t>(ws->ray_shard_cap, 1),
                                           (double)words[at + kCountBytes / sizeof(uint32_t)] /
                                               (double)std::max<size_t>(ws->shadow_shard_cap, 1)));
        }
      if (worst > 1.0 && per_launch > 1) {
        uint64_t finer = 2;
        while ((double)finer < worst * 1.05 && finer < per_launch) finer *= 2;
        sc->split *= finer;
        return enqueue_frame(sc, ws);  // (the shape is known now: no second probe)
      }
    }
  }

  if (n_pixels) {
    ResolveArgs r;
    std::memset(&r, 0, sizeof r);
    r.accum = ws->accum.as<double>();
    r.out = ws->d_out;
    r.n_pixels = (uint32_t)n_pixels;
    r.spp = spp;
    r.cfg = *config;
    launch_resolve((unsigned)((n_pixels + 255) / 256), stream, r);
    HIP_TRY(hipGetLastError());
    ws->launches++;
    ws->accum_clean = true;  // k_resolve zeroes what it reads
    ws->accum_clean_bytes = n_pixels * 3 * sizeof(double);
  }
  HIP_TRY(hipEventRecord(ws->ev_end, stream));
  HIP_TRY(hipMemcpyAsync(ws->pinned, ws->control.p, control_bytes, hipMemcpyDeviceToHost, stream));
  ws->pending = true;
  return MIRT_OK;
}

// Waits for the frame in `ws`.  If a queue shard overflowed, the frame is rendered again with finer
// primary launches until it fits.  Fills *stats.
int finish_frame(mirt_scene *sc, mirt_scene::WorkSet *ws, mirt_stats *stats) {
  if (!ws->pending) return fail(MIRT_ERR_INVALID, "no frame in flight in this slot");
  const int levels = ws->config.max_ray_depth;
  const size_t stats_bytes = COUNTER_SHARDS * sizeof(DCounters);
  uint64_t secondary = 0;
  for (;;) {
    HIP_TRY(hipStreamSynchronize(ws->stream));
    ws->pending = false;
    // did every queue shard hold what was appended to it?
    bool overflow = false;
    double worst = 0.0;  // most any shard was asked to hold, in capacities
    secondary = 0;
    const uint32_t *words = reinterpret_cast<const uint32_t *>(ws->pinned + stats_bytes);
    for (size_t c = 0; c < ws->chunks; ++c)
      for (int level = 0; level < levels; ++level)
        for (int s = 0; s < QUEUE_SHARDS; ++s) {
          const size_t at = ((c * (size_t)levels + (size_t)level) * 2) * (kCountBytes / sizeof(uint32_t)) +
                            (size_t)s * QUEUE_COUNT_STRIDE;
          const uint32_t rays = words[at], items = words[at + kCountBytes / sizeof(uint32_t)];
          if (rays > ws->ray_shard_cap || items > ws->shadow_shard_cap) overflow = true;
          worst = std::max(worst, std::max((double)rays / (double)std::max<size_t>(ws->ray_shard_cap, 1),
                                           (double)items / (double)std::max<size_t>(ws->shadow_shard_cap, 1)));
          secondary += rays;
        }
    if (!overflow) break;
    if (ws->per_launch == 1)
      return fail(MIRT_ERR_MEMORY,
                  "ray queues too small for this frame even one pixel block at a time; raise MIRT_RAY_QUEUE / "
                  "MIRT_SHADOW_QUEUE (entries)");
    // Render the frame again with finer primary launches.  The tails count what was asked for, dropped entries
    // included, so the split that fits is known in one go (levels below a level that dropped rays saw fewer
    // than they will: if that makes the estimate too small, the loop comes round again).
    uint64_t finer = 2;
    while ((double)finer < worst * 1.05 && finer < ws->per_launch) finer *= 2;
    sc->split *= finer;
    if (int rc = enqueue_frame(sc, ws)) return rc;
  }

  if (sc->record_progress && !ws->fixed && sc->progress.p) {
    sc->progress_host.resize((size_t)sc->progress_bands + 1);
    HIP_TRY(hipMemcpy(sc->progress_host.data(), sc->progress.p, sc->progress_host.size() * sizeof(unsigned long long),
                      hipMemcpyDeviceToHost));
  }
  if (stats) {
    const DCounters *host_counters = reinterpret_cast<const DCounters *>(ws->pinned);
    const FrameShape shape = frame_shape(ws->camera, ws->config, ws->width, ws->height, ws->spp, ws->tile_w, ws->tile_h,
                                         ws->rank, ws->world);
    uint64_t primary = 0;
    if (levels > 0 && shape.total_blocks > 0) {
      // samples of the pixels this rank owns
      uint64_t px = 0;
      for (int slot = 0; slot < shape.fr.n_slots; ++slot) {
        const int t = tile_of_slot(slot, ws->rank, shape.fr.tiles_x, ws->world, shape.fr.stagger);
        const int tx = t % shape.fr.tiles_x, ty = t / shape.fr.tiles_x;
        px += (uint64_t)std::min(ws->tile_w, ws->width - tx * ws->tile_w) *
              std::min(ws->tile_h, ws->height - ty * ws->tile_h);
      }
      primary = px * (uint64_t)ws->spp;
    }
    std::memset(stats, 0, sizeof *stats);
    DCounters c;
    std::memset(&c, 0, sizeof c);
    for (int k = 0; k < COUNTER_SHARDS; ++k) {
      c.shadow_segments += host_counters[k].shadow_segments;
      c.node_tests += host_counters[k].node_tests;
      c.prim_tests += host_counters[k].prim_tests;
      c.shadow_node_tests += host_counters[k].shadow_node_tests;
      c.shadow_prim_tests += host_counters[k].shadow_prim_tests;
    }
    stats->primary_rays = primary;
    stats->secondary_rays = secondary;
    stats->shadow_segments = c.shadow_segments;
    stats->node_tests = c.node_tests;
    stats->prim_tests = c.prim_tests;
    stats->shadow_node_tests = c.shadow_node_tests;
    stats->shadow_prim_tests = c.shadow_prim_tests;
    stats->launches = ws->launches;
    stats->frame_attempts = ws->attempts;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ws->ev_begin, ws->ev_end));
    stats->device_ms = ms;
    for (const Timed &t : ws->timed) {
      HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
      if (t.kind == 0) {
        stats->trace_kernel_ms += ms;
        stats->trace_kernel_launches++;
      } else {
        stats->shadow_kernel_ms += ms;
        stats->shadow_kernel_launches++;
      }
    }
  }
  return MIRT_OK;
}

// Records a frame in work set `slot` and enqueues it.
int begin_frame(mirt_scene *sc, int slot, const mirt_camera *camera, const mirt_config *config, int width, int height,
                int spp, int tile_w, int tile_h, int rank, int world, double *d_out, hipStream_t stream) {
  if (!sc || !camera || !config || !d_out) return fail(MIRT_ERR_INVALID, "NULL argument");
  if (slot < -1 || slot >= MIRT_SLOTS) return fail(MIRT_ERR_INVALID, "slot must be 0 .. %d", MIRT_SLOTS - 1);
  if (width <= 0 || height <= 0 || spp <= 0) return fail(MIRT_ERR_INVALID, "width, height and spp must be positive");
  if (tile_w <= 0 || tile_h <= 0 || world <= 0 || rank < 0 || rank >= world)
    return fail(MIRT_ERR_INVALID, "bad tile split: tile %dx%d rank %d of %d", tile_w, tile_h, rank, world);
  if ((double)width * height * spp > 4.0e9) return fail(MIRT_ERR_UNSUPPORTED, "more than 4e9 samples in one frame");
  mirt_scene::WorkSet *ws = slot < 0 ? &sc->fixed_work : &sc->work[slot];  // -1: the set that never allocates
  if (ws->pending) return fail(MIRT_ERR_INVALID, "slot %d still holds a frame: call mirt_render_tiles_end first", slot);
  HIP_TRY(hipSetDevice(sc->device));
  (void)hipGetLastError();  // an error some earlier, failed call left behind is not this frame's
  ws->camera = *camera;
  ws->config = *config;
  ws->width = width;
  ws->height = height;
  ws->spp = spp;
  ws->tile_w = tile_w;
  ws->tile_h = tile_h;
  ws->rank = rank;
  ws->world = world;
  ws->d_out = d_out;
  ws->stream = stream ? stream : sc->stream;
  if (!stream) {
    // NULL means "the engine's own stream", which does not synchronise with the null stream by itself:
    // make it wait for what the caller has already enqueued there (a fill of the output buffer, say)
    HIP_TRY(hipEventRecord(sc->null_stream_mark, nullptr));
    HIP_TRY(hipStreamWaitEvent(sc->stream, sc->null_stream_mark, 0));
  }
  ws->attempts = 0;
  return enqueue_frame(sc, ws);
}

// One frame (or this rank's tiles of it) into a device buffer, start to finish.
int render_device(mirt_scene *sc, const mirt_camera *camera, const mirt_config *config, int width, int height,
                  int spp, int tile_w, int tile_h, int rank, int world, double *d_out, hipStream_t stream,
                  mirt_stats *stats) {
  if (int rc = begin_frame(sc, 0, camera, config, width, height, spp, tile_w, tile_h, rank, world, d_out, stream))
    return rc;
  return finish_frame(sc, &sc->work[0], stats);
}

}  // namespace

MIRT_API int mirt_render_tiles_begin(mirt_scene *scene, const mirt_camera *camera, const mirt_config *config,
                                     int width, int height, int spp, int tile_w, int tile_h, int rank, int world,
                                     double *d_rgb_out, void *stream, int slot) {
  if (slot < 0 || slot >= MIRT_SLOTS) return fail(MIRT_ERR_INVALID, "slot must be 0 .. %d", MIRT_SLOTS - 1);
  return begin_frame(scene, slot, camera, config, width, height, spp, tile_w, tile_h, rank, world, d_rgb_out,
                     static_cast<hipStream_t>(stream));
}

MIRT_API int mirt_render_tiles_end(mirt_scene *scene, int slot, mirt_stats *stats) {
  if (!scene) return fail(MIRT_ERR_INVALID, "NULL scene");
  if (slot < 0 || slot >= MIRT_SLOTS) return fail(MIRT_ERR_INVALID, "slot must be 0 .. %d", MIRT_SLOTS - 1);
  HIP_TRY(hipSetDevice(scene->device));
  return finish_frame(scene, &scene->work[slot], stats);
}

MIRT_API int mirt_render_tiles(mirt_scene *scene, const mirt_camera *camera, const mirt_config *config, int width,
                               int height, int spp, int tile_w, int tile_h, int rank, int world, double *d_rgb_out,
                               void *stream, mirt_stats *stats) {
  return render_device(scene, camera, config, width, height, spp, tile_w, tile_h, rank, world, d_rgb_out,
                       static_cast<hipStream_t>(stream), stats);
}

MIRT_API int mirt_render_tiles_noalloc(mirt_scene *scene, const mirt_camera *camera, const mirt_config *config,
                                       int width, int height, int spp, int tile_w, int tile_h, int rank, int world,
                                       double *d_rgb_out, void *stream, mirt_stats *stats) {
  if (int rc = begin_frame(scene, -1, camera, config, width, height, spp, tile_w, tile_h, rank, world, d_rgb_out,
                           static_cast<hipStream_t>(stream)))
    return rc;
  return finish_frame(scene, &scene->fixed_work, stats);
}

MIRT_API uint64_t mirt_noalloc_scratch_bytes(void) { return MIRT_SCRATCH_BYTES; }

MIRT_API int mirt_scene_record_progress(mirt_scene *scene, int enable) {
  if (!scene) return fail(MIRT_ERR_INVALID, "NULL scene");
  scene->record_progress = enable != 0;
  if (!enable) scene->progress_host.clear();
  return MIRT_OK;
}

MIRT_API int mirt_scene_progress(mirt_scene *scene, double *band_ms, int max_bands) {
  if (!scene || (!band_ms && max_bands > 0)) return fail(MIRT_ERR_INVALID, "NULL argument");
  const std::vector<unsigned long long> &p = scene->progress_host;
  if (p.size() < 2) return 0;
  const int n = std::min<int>(max_bands, (int)p.size() - 1);
  unsigned long long start = p[0];
  if (start == ~0ull) {  // no start was noted: count from the earliest finish
    for (size_t i = 1; i < p.size(); ++i)
      if (p[i]) start = std::min(start, p[i]);
  }
  for (int i = 0; i < n; ++i)
    band_ms[i] = p[i + 1] >= start && p[i + 1] ? (double)(p[i + 1] - start) / scene->wall_clock_khz : 0.0;
  return (int)p.size() - 1;
}

// A finished frame from device memory to the caller's buffer, which is ordinary pageable memory (the
// reference's Image).  A plain hipMemcpy into pageable memory runs at about 9 GB/s here: 5.5 ms for a
// 1080p frame that took 6 ms to render.  So the frame goes to pinned staging memory in a few pieces
// (DMA at link speed), and each piece is copied on by a thread of its own as soon as it has arrived.
// Not part of the ABI; also used by the device group (multi.cpp).  `src` must be complete: the copy
// runs on the scene's own stream.
int mirt_copy_frame_out(mirt_scene *sc, void *dst, const void *src, size_t bytes) {
  constexpr int kPieces = mirt_scene::kStagePieces;
  if (bytes < ((size_t)4 << 20)) {
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return MIRT_OK;
  }
  if (sc->stage_bytes < bytes) {
    if (sc->stage) (void)hipHostFree(sc->stage);
    sc->stage = nullptr;
    sc->stage_bytes = 0;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {  // no pinned memory to be had: the plain way
      (void)hipGetLastError();
      HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
      return MIRT_OK;
    }
    sc->stage = static_cast<char *>(p);
    sc->stage_bytes = bytes;
  }
  for (hipEvent_t &e : sc->stage_done)
    if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const size_t piece = ((bytes + kPieces - 1) / kPieces + 4095) & ~(size_t)4095;
  size_t off[kPieces + 1];
  for (int k = 0; k <= kPieces; ++k) off[k] = std::min(bytes, (size_t)k * piece);
  for (int k = 0; k < kPieces; ++k) {
    if (off[k + 1] > off[k])
      HIP_TRY(hipMemcpyAsync(sc->stage + off[k], static_cast<const char *>(src) + off[k], off[k + 1] - off[k],
                             hipMemcpyDeviceToHost, sc->stream));
    HIP_TRY(hipEventRecord(sc->stage_done[k], sc->stream));
  }
  hipError_t err[kPieces] = {};
  auto carry = [&](int k) {
    err[k] = hipEventSynchronize(sc->stage_done[k]);
    if (err[k] == hipSuccess && off[k + 1] > off[k])
      std::memcpy(static_cast<char *>(dst) + off[k], sc->stage + off[k], off[k + 1] - off[k]);
  };
  std::thread helpers[kPieces - 1];
  int started = 0;
  try {
    for (int k = 1; k < kPieces; ++k) {
      helpers[k - 1] = std::thread(carry, k);
      started = k;
    }
  } catch (const std::exception &) {  // no more threads to be had: this one carries the rest
  }
  carry(0);
  for (int k = started + 1; k < kPieces; ++k) carry(k);
  for (int k = 1; k <= started; ++k) helpers[k - 1].join();
  for (int k = 0; k < kPieces; ++k)
    if (err[k] != hipSuccess) return fail(MIRT_ERR_DEVICE, "frame read-back: %s", hipGetErrorString(err[k]));
  return MIRT_OK;
}

MIRT_API int mirt_render(mirt_scene *scene, const mirt_camera *camera, const mirt_config *config, int width,
                         int height, int spp, double *rgb_out, mirt_stats *stats) {
  if (!scene || !rgb_out) return fail(MIRT_ERR_INVALID, "NULL argument");
  if (width <= 0 || height <= 0) return fail(MIRT_ERR_INVALID, "width and height must be positive");
  HIP_TRY(hipSetDevice(scene->device));
  const size_t bytes = (size_t)width * height * 3 * sizeof(double);
  HIP_TRY(scene->out.reserve(bytes));
  if (int rc = render_device(scene, camera, config, width, height, spp, width, height, 0, 1, scene->out.as<double>(),
                             nullptr, stats))
    return rc;
  return mirt_copy_frame_out(scene, rgb_out, scene->out.p, bytes);
}

MIRT_API int mirt_trace_ids(mirt_scene *scene, const mirt_camera *camera, int width, int height, int32_t *prim_id,
                            double *t, double *normal, double *uv) {
  if (!scene || !camera || !prim_id) return fail(MIRT_ERR_INVALID, "NULL argument");
  if (width <= 0 || height <= 0) return fail(MIRT_ERR_INVALID, "width and height must be positive");
  (void)hipGetLastError();  // an error some earlier, failed call left behind is not this call's
  HIP_TRY(hipSetDevice(scene->device));
  const size_t n = (size_t)width * height;
  // layout: ids | t | normal | uv
  const size_t off_t = (n * sizeof(int32_t) + 63) & ~(size_t)63;
  const size_t off_n = off_t + n * sizeof(double);
  const size_t off_uv = off_n + 3 * n * sizeof(double);
  HIP_TRY(scene->ids.reserve(off_uv + 2 * n * sizeof(double)));
  char *base = scene->ids.as<char>();
  mirt_camera pin = *camera;
  pin.dof_enabled = 0;
  IdsArgs a;
  std::memset(&a, 0, sizeof a);
  a.sc = scene->d;
  a.cam = device_camera(pin, width, height);
  a.width = width;
  a.height = height;
  a.prim_id = reinterpret_cast<int32_t *>(base);
  a.t = reinterpret_cast<double *>(base + off_t);
  a.normal = reinterpret_cast<double *>(base + off_n);
  a.uv = reinterpret_cast<double *>(base + off_uv);
  const unsigned grid = (unsigned)(((width + 15) / 16) * ((height + 15) / 16));
  launch_trace_ids(scene->walk, grid, scene->stream, a);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(scene->stream));
  HIP_TRY(hipMemcpy(prim_id, a.prim_id, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (t) HIP_TRY(hipMemcpy(t, a.t, n * sizeof(double), hipMemcpyDeviceToHost));
  if (normal) HIP_TRY(hipMemcpy(normal, a.normal, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
  if (uv) HIP_TRY(hipMemcpy(uv, a.uv, 2 * n * sizeof(double), hipMemcpyDeviceToHost));
  return MIRT_OK;
}

MIRT_API int mirt_trace_rays(mirt_scene *scene, int64_t n, const double *origin, const double *dir, const double *time,
                             double t_min, const double *t_max, int32_t *prim_id, double *t, double *normal) {
  if (!scene || !origin || !dir || !prim_id) return fail(MIRT_ERR_INVALID, "NULL argument");
  if (n < 0 || n > 0x7fffffff) return fail(MIRT_ERR_INVALID, "ray count out of range");
  if (n == 0) return MIRT_OK;
  (void)hipGetLastError();  // an error some earlier, failed call left behind is not this call's
  HIP_TRY(hipSetDevice(scene->device));
  const size_t N = (size_t)n;
  // layout: origin 3N | dir 3N | time N | t_max N | t N | normal 3N | ids N
  HIP_TRY(scene->ids.reserve(N * (12 * sizeof(double) + sizeof(int32_t)) + 64));
  double *base = scene->ids.as<double>();
  RaysArgs a;
  std::memset(&a, 0, sizeof a);
  a.sc = scene->d;
  double *d_o = base, *d_d = base + 3 * N, *d_time = base + 6 * N, *d_tmax = base + 7 * N;
  a.t = base + 8 * N;
  a.normal = base + 9 * N;
  a.prim_id = reinterpret_cast<int32_t *>(base + 12 * N);
  HIP_TRY(hipMemcpy(d_o, origin, 3 * N * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_d, dir, 3 * N * sizeof(double), hipMemcpyHostToDevice));
  if (time) HIP_TRY(hipMemcpy(d_time, time, N * sizeof(double), hipMemcpyHostToDevice));
  if (t_max) HIP_TRY(hipMemcpy(d_tmax, t_max, N * sizeof(double), hipMemcpyHostToDevice));
  a.origin = d_o;
  a.dir = d_d;
  a.time = time ? d_time : nullptr;
  a.t_max = t_max ? d_tmax : nullptr;
  a.t_min = t_min;
  a.n = (uint32_t)N;
  launch_trace_rays(scene->walk, (unsigned)((N + 255) / 256), scene->stream, a);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(scene->stream));
  HIP_TRY(hipMemcpy(prim_id, a.prim_id, N * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (t) HIP_TRY(hipMemcpy(t, a.t, N * sizeof(double), hipMemcpyDeviceToHost));
  if (normal) HIP_TRY(hipMemcpy(normal, a.normal, 3 * N * sizeof(double), hipMemcpyDeviceToHost));
  return MIRT_OK;
}
