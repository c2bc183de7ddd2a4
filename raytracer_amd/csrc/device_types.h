// Device-side data layout of the engine (HBM-resident, read-only during a frame) and the records
// that flow between kernels.  Everything is IEEE double, as in the reference renderer.
//
//   DNode    64 B   BVH node in depth-first order with a skip link (index of the first node after
//                   this node's subtree): the reference's fixed left-then-right recursion
//                   (Code/Geometry/Bvh.cpp:306-333) becomes a stack-free walk  i -> i+1 | skip[i].
//   DPrim   256 B   one primitive: kind, flags, the static object<->world maps (for a moving
//                   primitive: the maps at shutter time 0.5, which is the time of every shadow ray,
//                   Code/Core/Ray.h:15, and of every ray when motion blur is off) and the shape
//                   parameters.  Planes keep their pre-computed plane equation in the same slots.
//   DMotion 384 B   motion-blur data of a moving primitive: the TRS decomposition of matrix_t0 /
//                   matrix_t1 done once on the host (Code/Math/Transform.cpp:145-206), so that a ray
//                   only lerps, slerps and composes (Transform.cpp:239-256).
#pragma once

#include <stdint.h>

#include "mirt.h"

enum : int32_t { MOTION_NONE = 0, MOTION_TRANSLATE = 1, MOTION_GENERAL = 2 };

struct alignas(64) DNode {
  double bmin[3];
  double bmax[3];
  int32_t skip;   // next node in depth-first order after this subtree
  int32_t first;  // leaf: first entry in leaf_prims; inner node: -1
  int32_t count;  // leaf: number of primitives; inner node: index of the second child (the first is this + 1)
  int32_t order;  // inner node: bits 0-1 axis along which the children lie furthest apart, bit 2: second child is the lower one; leaf: leaf_tagged[first] (0 when empty)
};
static_assert(sizeof(DNode) == 64, "DNode must be one 64-byte record");

// BVH walks (see pipeline.h)
enum : int32_t { WALK_PACKET = 0, WALK_LANE = 1, WALK_ORDERED = 2, WALK_WIDE = 3 };
// leaf_tagged: leaf_prims with the primitive's kind in the top bits, for the walk that batches by kind,
// and one bit that says "transparency exactly 1": a shadow segment passes such a surface unattenuated
// (Raytracer.cpp:190-198), so the any-hit form of the shadow walk does not test it at all
#define LEAF_KIND_SHIFT 28
#define LEAF_CLEAR_BIT 0x08000000
#define LEAF_PRIM_MASK 0x07ffffff

// The wide walk's view of the same tree: the binary nodes collapsed two levels at a time.
//   DWide  128 B  four slots: the grandchildren of a binary inner node (a child that is a leaf fills
//                 one slot of its pair and leaves the other empty), boxes rounded outwards to float;
//                 child >= 0: wide node, < 0: ~index of a DLeaf; an empty slot has an inverted box,
//                 which no ray hits.  order: 3 bits for the binary node itself, 3 for its first
//                 child, 3 for its second (DNode::order of each).
//   DLeaf   64 B  a binary leaf: its exact box and up to four primitives (index | kind <<
//                 LEAF_KIND_SHIFT, -1 = none); a longer leaf sets prim[3] = LEAF_INDIRECT and names
//                 leaf_tagged[prim[0] .. prim[0] + prim[1]).
struct alignas(128) DWide {
  float lo[4][3];
  float hi[4][3];
  int32_t child[4];
  uint32_t order;
  uint32_t pad[3];
};
static_assert(sizeof(DWide) == 128, "DWide must be one 128-byte line");
struct alignas(64) DLeaf {
  double bmin[3];
  double bmax[3];
  int32_t prim[4];
};
static_assert(sizeof(DLeaf) == 64, "DLeaf must be 64 bytes");
#define WIDE_NONE INT32_MIN
#define LEAF_INDIRECT (INT32_MIN + 1)
#define WIDE_STACK_LDS 12   // the wide walk's stack entries per lane kept in LDS; deeper ones go to the lane's private memory
#define WIDE_STACK_DEEP 60  // per-lane stack entries beyond those in LDS (a visit pops one entry and pushes at most four)
#define WALK_STACK 24  // entries of the ordered walk's per-lane stack: the tree may be this deep at most

struct alignas(64) DPrim {
  int32_t type;
  int32_t visible;
  int32_t motion;    // MOTION_*
  int32_t material;
  int32_t motion_index;  // into DScene::motion
  int32_t plane_flags;   // plane: bit a set when extent along axis a exceeds the 1e-6 tolerance
  int32_t pad0, pad1;
  // transformed kinds: xf[0..11] = world_to_object rows 0-2 (3x4), xf[12..23] = object_to_world
  // plane: xf[0..2] first point, [3..5] unit normal, [6..8] min corner, [9..11] max corner,
  //        [12..14] unit edge 1, [15] |edge 1|, [16..18] unit edge 2, [19] |edge 2|
  double xf[24];
  double param[4];
};
static_assert(sizeof(DPrim) == 256, "DPrim must be 256 bytes");

struct alignas(64) DMotion {
  double t0[3], dt[3];  // translation at t=0 and its change over the shutter
  double s0[3], ds[3];  // scale likewise
  double q0[4], dq[4];  // unit start quaternion (w,x,y,z); end - start (lerp path) or the end quaternion (slerp path)
  double dot;           // cosine between the two (after the shorter-arc sign flip)
  double theta0;        // acos(dot)
  double inv_sin0;      // 1 / sin(theta0)
  int32_t lerp;         // dot > 0.9995: normalised lerp (Quaternion.h:88-96), else slerp
  int32_t pad;
  // MOTION_TRANSLATE only: constant rotation*scale block (row-major 3x3) and its inverse
  double m[9];
  double minv[9];
};
static_assert(sizeof(DMotion) == 384, "DMotion layout");

struct DTexture {
  const uint16_t *rgb;
  int32_t width, height;
  double inv_max;  // not used for the value itself (division is kept), only as a presence marker
  int32_t max_value, pad;
};

struct DScene {
  const DNode *nodes;
  const int32_t *leaf_prims;
  const int32_t *leaf_tagged;  // leaf_prims[i] | kind << LEAF_KIND_SHIFT
  const DWide *wide;           // the wide walk's nodes (root: 0) and leaves
  const DLeaf *leaves;
  const DPrim *prims;
  const DMotion *motion;
  const mirt_material *materials;
  const double *prim_transparency;  // materials[prims[i].material].transparency, for shadow rays
  const mirt_light *lights;
  const DTexture *textures;
  int32_t n_nodes, n_prims, n_lights, n_textures;
  double background[3];  // background_color * background_strength (Raytracer.cpp:472)
  int32_t kinds;         // bit k: some primitive is of kind k (MIRT_SPHERE ... MIRT_CONE)
  int32_t any_motion;    // some primitive has motion data
  int32_t any_maps;      // some material samples a texture, normal map or bump map
  int32_t max_children;  // most rays one hit can spawn, given the materials (and glossy_samples, added per frame)
  // every primitive is either opaque (transparency <= 0) or perfectly clear (== 1): a shadow stratum is
  // then 1 if any opaque primitive lies between the point and the light and 0 otherwise, and the shadow
  // walk may stop at the first opaque hit it finds instead of chaining closest hits
  int32_t shadow_any_hit;
  int32_t pad_;
  // The wide walk's guard against surfaces that coincide (see wide_step): tie_hi = 1 + band widens the distance
  // a primitive is tested against, tie_lo = (1 - band) / (1 + band) marks a hit that lands within the band of the
  // one it replaces; such rays are walked again in the reference's order.  Both 1: no band (no binary tree to
  // walk again, or switched off).
  double tie_hi, tie_lo;
};

// Camera with the basis of Camera::get_ray (Camera.cpp:32-58) evaluated once on the host.
struct DCamera {
  double location[3];
  double gaze[3];  // as parsed, un-normalised (used by the focus-plane formula, Camera.cpp:83)
  double w[3], u[3], v[3];
  double viewport_w, viewport_h, image_plane_dist;
  double focus_distance, aperture_radius;
  int32_t dof;
  int32_t pad;
};

// Per-frame constants.
struct DFrame {
  DCamera cam;
  mirt_config cfg;
  int32_t width, height, spp;
  int32_t tile_w, tile_h, tiles_x, tiles_y, rank, world;
  int32_t stagger;             // tile ownership is staggered row by row (tile_stagger)
  int32_t blocks_x, blocks_y;  // 16x16-pixel thread blocks per tile
  int32_t n_slots;             // tiles owned by this rank
  uint32_t plane;              // pixels of this rank's tile buffer: the accumulator is three planes of that many doubles
  const struct DLightFrame *light_frame;  // per light, per frame (depends on cfg.shadow_samples)
};

// Shadow sampling of one light for this frame (Raytracer.cpp:146-158).
struct DLightFrame {
  uint32_t dim;   // first random dimension of this light's strata (mirt_rng.h, MIRT_DIM_SHADOW)
  int32_t side;   // floor(sqrt(samples)), at least 1; side*side strata are taken
  double inv_n;   // 1 / (side*side)
  // area lights: the basis Light::sample_point (Light.h:44-52) builds on every call, built once
  double right[3], up[3];
};

// Which rank renders which tile.  Tile (tx, ty) of the row-major tile grid belongs to rank
// (ty * tiles_x + tx) % world -- unless tiles_x is a multiple of world, where that would give every rank
// whole columns of tiles (a systematic imbalance on anything with vertical structure).  Then the owner is
// (tx + k * ty) % world with k = round(sqrt(world)): the tiles of one rank form a lattice whose points are
// about sqrt(world) tiles apart in every direction (k = 1, plain diagonals, leaves them 1.4 tiles apart
// along the diagonal and `world` apart across it, and a feature a few tiles wide then lands on some
// ranks more than on others).  A rank's buffer holds its tiles in row-major order.
#ifdef __HIPCC__
#define MIRT_HD __host__ __device__
#else
#define MIRT_HD
#endif
// 0: the row-major rule; otherwise the multiplier k of the lattice rule
MIRT_HD static inline int tile_stagger(int tiles_x, int world) {
  if (world <= 1 || tiles_x % world != 0) return 0;
  int k = 1;
  while ((2 * k + 1) * (2 * k + 1) <= 4 * world) ++k;  // round(sqrt(world))
  return k;
}
MIRT_HD static inline int tile_owner(int tx, int ty, int tiles_x, int world, int stagger) {
  return stagger ? (tx + stagger * ty) % world : (ty * tiles_x + tx) % world;
}
// the row-major index of the tile in slot `slot` of rank `rank`'s buffer
MIRT_HD static inline int tile_of_slot(int slot, int rank, int tiles_x, int world, int stagger) {
  if (!stagger) return rank + slot * world;
  const int per_row = tiles_x / world, ty = slot / per_row, j = slot % per_row;
  return ty * tiles_x + ((rank - stagger * ty) % world + world) % world + j * world;
}
// the slot of tile (tx, ty) in its owner's buffer
MIRT_HD static inline int tile_slot(int tx, int ty, int tiles_x, int world, int stagger) {
  return stagger ? ty * (tiles_x / world) + tx / world : (ty * tiles_x + tx) / world;
}

// Ray queue, structure of arrays; one queue per ray-tree depth.
// Queues are cut into QUEUE_SHARDS independent shards, each with its own tail counter on its own
// 256-byte line.  A producer block appends to shard blockIdx % QUEUE_SHARDS, so the one atomic a wave
// issues per append contends with 1/QUEUE_SHARDS of the other waves only: appends to a single tail
// word serialise at about 11 ns each on this chip, which for the ~10^6 waves of a 1080p x 16 spp launch
// would cost more than the tracing itself.
#define QUEUE_SHARDS 32
#define QUEUE_COUNT_STRIDE 64  // uint32 words between two shards' tail counters

struct DRayQueue {
  double *ox, *oy, *oz, *dx, *dy, *dz, *time, *wr, *wg, *wb;
  uint64_t *key;
  uint32_t *pixel;
  uint32_t shard_cap;  // entries per shard; shard s owns slots [s * shard_cap, (s + 1) * shard_cap)
  uint32_t pad;
};
// A queue's tail counters live apart from its storage (QUEUE_SHARDS words, QUEUE_COUNT_STRIDE apart,
// one block per launch that fills the queue), so that the host can read all of a frame's counts at
// once when the frame is over.  Consumer kernels read the counts themselves: no launch waits for
// the host to learn how much the launch before it produced.

// Shadow work item: one (hit, light) pair; 64 bytes.
struct alignas(16) DShadowItem {
  double p[3];   // shaded point
  double c[3];   // unshadowed contribution of this light to the pixel, already divided by the strata count
  uint64_t key;  // random key of the ray-tree node that owns the hit
  uint32_t pixel;
  uint32_t light;
};
static_assert(sizeof(DShadowItem) == 64, "DShadowItem must be 64 bytes");

// Closest hit of one ray, as k_closest_stream leaves it for k_shade.
struct alignas(16) DHit {
  int32_t prim, pad;  // -1: no hit
  double t;
  double p[3], n[3];  // world hit point; outward world normal before normalisation (planes: unit)
};
static_assert(sizeof(DHit) == 64, "DHit must be 64 bytes");
struct alignas(16) DHitUV {
  double u, v, tan[3], pad;
};
static_assert(sizeof(DHitUV) == 48, "DHitUV must be 48 bytes");

// Statistics counters, replicated COUNTER_SHARDS times (one 64-byte line each) for the same reason
// as the queue tails; the host adds the replicas up after the frame.
#define COUNTER_SHARDS 256
struct alignas(64) DCounters {
  unsigned long long primary, secondary, shadow_segments, node_tests, prim_tests, shadow_node_tests,
      shadow_prim_tests, pad;
};

// ---- kernel argument blocks (passed by value) --------------------------------------------------
struct TraceArgs {
  DScene sc;
  DFrame fr;
  DRayQueue in;              // unused by the primary launch
  const uint32_t *in_count;  // tail counters of `in` as its producer left them
  DRayQueue out;
  uint32_t *out_count;       // NULL when the rays of this depth spawn nothing
  DShadowItem *shadow;
  uint32_t *shadow_count;
  uint32_t shadow_shard_cap;
  uint32_t shadow_groups;    // > 1: shadow items are queued by light, light li to shards [li * shadow_per, (li + 1) * shadow_per)
  uint32_t shadow_per;
  int32_t depth;             // depth of the rays this launch traces
  double *accum;             // this rank's tile buffer: a red, a green and a blue plane of fr.plane doubles
  DCounters *counters;       // COUNTER_SHARDS replicas
  // PRIMARY: this launch renders n 16x16 blocks: the runs of `group` consecutive blocks numbered first, first +
  // stride, first + 2 stride, ... (blocks >= total do not exist).  The primary launches of a frame interleave
  // their runs (stride = number of launches), so that each launch is a sample of the whole image and asks the
  // same of the queues: a launch over the image's top rows alone says nothing about what the floor below spawns.
  uint32_t first;
  uint32_t n;
  uint32_t stride;
  uint32_t group;
  uint32_t total;
  uint32_t pad_;
  unsigned long long *progress;  // PRIMARY, optional: [0] earliest start, [1 + r] latest finish of 16-pixel row band r (wall clock)
  uint32_t *grab;            // k_closest_stream: the next ray number nobody has taken yet (zero at launch)
  DHit *hits;                // k_closest_stream -> k_shade, one per ray number
  DHitUV *hits_uv;
};

struct ShadowArgs {
  DScene sc;
  DFrame fr;
  const DShadowItem *items;
  const uint32_t *count;
  uint32_t shard_cap;
  uint32_t pad;
  double *accum;
  DCounters *counters;
  uint32_t *grab;  // k_shadow_stream: the next item number nobody has taken yet (zero at launch)
};

struct ResolveArgs {
  double *accum;  // read, then zeroed for the next frame
  double *out;
  uint32_t n_pixels;
  int32_t spp;
  mirt_config cfg;
};

struct IdsArgs {
  DScene sc;
  DCamera cam;
  int32_t width, height;
  int32_t *prim_id;
  double *t, *normal, *uv;
};

struct RaysArgs {
  DScene sc;
  const double *origin, *dir, *time, *t_max;  // 3n, 3n, n or NULL (0.5), n or NULL (DBL_MAX)
  double t_min;
  uint32_t n;
  int32_t *prim_id;
  double *t, *normal;
};

// Tiles of a group's members, gathered side by side on the root device, into the frame (multi.cpp).
struct DeinterleaveArgs {
  const double *gathered;
  double *frame;
  int32_t width, height, tile, tiles_x, world, stagger;
  uint64_t offset[MIRT_GROUP_MAX];  // in doubles: where member r's tiles start in `gathered`
};
void launch_deinterleave(unsigned grid, void *stream, const DeinterleaveArgs &a);

// Launchers defined in kernels.hip (the only translation unit compiled for the GPU).
// walk: WALK_* (wave-coherent, per lane in the reference's order, per lane nearer child first, 4-wide and batched);
// uv: the scene samples image maps, so hits carry texture coordinates and a tangent frame;
// motion: some ray of the launch may be at a shutter time other than 0.5.
void launch_trace(int walk, bool primary, bool uv, bool motion, unsigned grid, void *stream, const TraceArgs &a);
// strata: some light has more than one; point: no light is an area light
void launch_shadow(int walk, bool strata, bool any_hit, bool point, unsigned grid, void *stream, const ShadowArgs &a);
// wide walk, resident grid; any_hit: segments end at the first opaque hit (DScene::shadow_any_hit)
void launch_shadow_stream(bool any_hit, unsigned grid, void *stream, const ShadowArgs &a);
// wide walk, two launches: closest hits by a resident grid, then shading with one lane per ray
void launch_closest_stream(bool primary, bool uv, bool motion, unsigned grid, void *stream, const TraceArgs &a);
void launch_shade(bool primary, bool uv, unsigned grid, void *stream, const TraceArgs &a);
void launch_resolve(unsigned grid, void *stream, const ResolveArgs &a);
void launch_trace_ids(int walk, unsigned grid, void *stream, const IdsArgs &a);
void launch_trace_rays(int walk, unsigned grid, void *stream, const RaysArgs &a);

// ---- device-side BVH construction (bvh_build.h) ---------------------------------------------------
// A binary radix tree over the primitives sorted by the Morton code of their box centres (internal
// node i < n - 1, leaf k = sorted position k; a node reference is >= 0 for an internal node and ~k for
// a leaf), its boxes fitted bottom-up, emitted top-down as the DWide / DLeaf arrays the wide walk reads.
#define BUILD_REF_NONE INT32_MIN
#define BUILD_SORT_ITEMS 16  // keys per lane of a sorting wave (one wave per block)
#define BUILD_SORT_TILE (64 * BUILD_SORT_ITEMS)
struct BuildArgs {
  const DPrim *prims;
  const double *motion_ends;   // per moving primitive (DPrim::motion_index): object_to_world rows 0-2 of matrix_t0, then of matrix_t1
  const double *transparency;  // per primitive
  uint32_t n, leaf_max;        // primitives; most primitives of a leaf (1 .. 4)
  double *pbox;                // 6 per primitive: world box (min, max)
  unsigned long long *bounds;  // 6 words: min x, y, z and max x, y, z of the box centres, in an order-preserving encoding
  uint64_t *keys, *keys_alt;   // Morton codes; the sort ping-pongs between the two
  uint32_t *ids, *ids_alt;     // the primitive at each sorted position
  uint32_t *hist;              // 256 digit counts per sorting block, digit-major
  uint32_t sort_blocks, shift;
  int32_t *child;              // 2 per internal node
  int32_t *parent;             // of the n - 1 internal nodes, then of the n leaves
  uint32_t *range;             // 2 per internal node: first and last sorted position it covers
  double *nbox;                // 6 per internal node
  uint32_t *visits;            // per internal node: children fitted so far
  DWide *wide;
  DLeaf *leaves;
  int32_t *wide_src;           // per wide node: the binary node it was collapsed from
  int32_t *leaf_src;           // per DLeaf: the binary node (or leaf) it stands for
  uint32_t *counters;          // [0] wide nodes allocated, [1] DLeafs allocated
  uint32_t level_begin, level_end;  // k_build_emit: the wide nodes of this level
};
void launch_build_boxes(void *stream, const BuildArgs &a);
void launch_build_keys(void *stream, const BuildArgs &a);
void launch_build_sort_pass(void *stream, const BuildArgs &a);  // one 8-bit digit at a.shift: keys/ids -> keys_alt/ids_alt
void launch_build_tree(void *stream, const BuildArgs &a);
void launch_build_fit(void *stream, const BuildArgs &a);
void launch_build_emit(void *stream, const BuildArgs &a);
void launch_build_refresh(void *stream, const BuildArgs &a, uint32_t n_wide, uint32_t n_leaves);  // refit: boxes and order codes again

// Size of the per-GPU static array behind mirt_render_tiles_noalloc (kernels.hip), and the device
// address of the current device's copy of it (0, or a hipError_t).
#define MIRT_SCRATCH_BYTES (64u << 20)
int scratch_address(int device, void **out);
