/* mirt.h -- C ABI of the MI355X (gfx950) ray-tracing engine.
 *
 * The reference renderer has no FFI: its render hot path is entered through one free function,
 *     Image render_scene_bvh_antialiased(const Scene&, const Camera&, int w, int h, int spp, int)
 *     (Code/Render/Raytracer.h:65-67, defined Raytracer.cpp:612-621, called from Main.cpp:196-198),
 * i.e. `Raytracer rt(scene); rt.render(camera, w, h, spp)` (Raytracer.h:27,33-34).  The entry points
 * below are what a host implementation of that function binds: the host flattens its Scene into
 * the plain arrays of mirt_scene_desc, hands them over once, and asks for frames.
 *
 * Conventions: plain C structs, no C++ types, no exceptions across the boundary.  Every function
 * returns MIRT_OK (0) or a negative mirt_status and leaves a message for mirt_last_error() (kept per
 * calling thread).  Calls block until the result is complete.  A mirt_scene handle owns all device
 * memory it allocates and may be used by one thread at a time.  All floating-point data is IEEE
 * double, exactly as in the reference (Code/Math/Vector.h:111-119).
 */
#ifndef MIRT_H
#define MIRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRT_ABI_VERSION 1

typedef enum mirt_status {
  MIRT_OK = 0,
  MIRT_ERR_INVALID = -1,   /* bad argument or inconsistent scene description */
  MIRT_ERR_DEVICE = -2,    /* HIP runtime error (no device, launch failure, ...) */
  MIRT_ERR_MEMORY = -3,    /* host or device allocation failed */
  MIRT_ERR_UNSUPPORTED = -4
} mirt_status;

/* Primitive kinds.  A primitive's index in mirt_scene_desc.prims IS its identity: the reference
 * numbers objects spheres, cubes, planes, toruses, cylinders, cones in that order (Bvh.cpp:244-304,
 * Raytracer.cpp:29-36) and the flattened array keeps that order. */
enum { MIRT_SPHERE = 0, MIRT_CUBE = 1, MIRT_PLANE = 2, MIRT_TORUS = 3, MIRT_CYLINDER = 4, MIRT_CONE = 5 };

/* One BVH node, 64 bytes (BVHNode, Bvh.h:18-27).  Nodes are stored in the depth-first order of
 * the reference's recursion (intersect_bvh, Bvh.cpp:306-333: node, left subtree, right subtree), so
 * the left child of inner node i is i + 1.  A leaf has left = right = -1 and names
 * leaf_prims[first .. first + count); an inner node has count = 0. */
typedef struct mirt_node {
  double bmin[3];
  double bmax[3];
  int32_t left;
  int32_t right;
  int32_t first;
  int32_t count;
} mirt_node;

/* One primitive (Shape, Shape.h:8-21, plus the per-kind fields of Sphere.h/Cube.h/Plane.h/Torus.h/
 * Cylinder.h/Cone.h).  Matrices are row-major 4x4 (Mat4, Transform.h:12-13). */
typedef struct mirt_prim {
  int32_t type;       /* MIRT_SPHERE ... MIRT_CONE */
  int32_t visible;    /* Shape::visible (Bvh.cpp:255-278) */
  int32_t has_motion; /* Shape::has_motion */
  int32_t material;   /* index into materials */
  double o2w[16];     /* cached_transform.object_to_world (SceneLoader.cpp:868-889) */
  double w2o[16];     /* cached_transform.world_to_object */
  double m0[16];      /* start_transform (matrix_t0) */
  double m1[16];      /* end_transform (matrix_t1) */
  double param[4];    /* torus: major, minor radius; cylinder, cone: radius, depth; cube: scale x, y, z */
  int32_t first_point; /* plane: range in plane_points (3 doubles per point) */
  int32_t n_points;
} mirt_prim;

/* Material (Material.h:7-49), with texture names resolved to indices into `textures`.  An index is
 * -1 when the name is empty or the image is not loaded (the reference loads the maps of spheres,
 * cubes and planes only: Raytracer.cpp:94-108). */
typedef struct mirt_material {
  double diffuse[3];
  double specular[3];
  double ambient[3];
  double emission[3];
  double shininess;
  double glossiness;
  double reflectivity;
  double transparency;
  double refractive_index;
  double emission_strength;
  double bump_strength;
  int32_t has_texture; /* Material::has_texture */
  int32_t texture;     /* texture_file */
  int32_t normal_map;
  int32_t bump_map;
} mirt_material;

/* Light (Light.h:8-34).  Every kind shades as a point light at `location` (Raytracer.cpp:266-271);
 * only AREA lights sample their surface for visibility (Light.h:38-69). */
enum { MIRT_LIGHT_POINT = 0, MIRT_LIGHT_AREA = 1 };
enum { MIRT_AREA_RECT = 0, MIRT_AREA_DISK = 1 }; /* SQUARE/RECTANGLE vs anything else (Light.h:59-68) */
typedef struct mirt_light {
  double location[3];
  double color[3];
  double normal[3];
  double intensity;
  double area_size[2];
  int32_t type;
  int32_t area_shape;
  int32_t samples;
  int32_t pad_;
} mirt_light;

/* Texture image (Image, Image.h:10-15): width * height RGB triples, row 0 first, as read from the
 * PPM; a texel's value is rgb / max_value (Image.cpp:118-120). */
typedef struct mirt_texture {
  int32_t width;
  int32_t height;
  int32_t max_value;
  int32_t pad_;
  const uint16_t *rgb;
} mirt_texture;

typedef struct mirt_scene_desc {
  int32_t n_nodes;
  int32_t n_leaf_prims;
  int32_t n_prims;
  int32_t n_plane_points;
  int32_t n_materials;
  int32_t n_lights;
  int32_t n_textures;
  int32_t pad_;
  const mirt_node *nodes;
  const int32_t *leaf_prims;
  const mirt_prim *prims;
  const double *plane_points;
  const mirt_material *materials;
  const mirt_light *lights;
  const mirt_texture *textures;
  double background_color[3]; /* SceneSettings (Scene.h:18-19) */
  double background_strength;
} mirt_scene_desc;

/* Camera (Camera.h:10-30), the fields Camera::get_ray reads (Camera.cpp:10-111). */
typedef struct mirt_camera {
  double location[3];
  double gaze[3];
  double up[3];
  double focal_length;
  double sensor_height;
  double focus_distance;
  double aperture_fstop;
  int32_t dof_enabled;
  int32_t pad_;
} mirt_camera;

enum { MIRT_TONEMAP_NONE = 0, MIRT_TONEMAP_REINHARD = 1, MIRT_TONEMAP_EXPOSURE = 2 };
enum {
  MIRT_SAMPLING_JITTER = 0, /* u = x + rnd, v = y + rnd, time = rnd or 0.5 (Raytracer.cpp:533-538) */
  MIRT_SAMPLING_CENTRE = 1  /* u = x + 0.5, v = y + 0.5, time = 0.5: the deterministic parity mode */
};

/* Snapshot of the RenderConfig fields the render path reads (Config.h:11-168), plus the engine's
 * own seed and sampling mode. */
typedef struct mirt_config {
  int32_t max_ray_depth;       /* Raytracer.cpp:297 */
  int32_t shadow_samples;      /* Raytracer.cpp:146-149 */
  int32_t glossy_samples;      /* Raytracer.cpp:331-337 */
  int32_t enable_motion_blur;  /* Raytracer.cpp:538 */
  int32_t enable_textures;     /* Raytracer.cpp:221,229,250,391 */
  int32_t use_adaptive_epsilon;/* Raytracer.cpp:172,323,427 */
  int32_t tone_mapping_mode;   /* Raytracer.cpp:547-559 */
  int32_t enable_gamma_correction; /* Raytracer.cpp:563 */
  double light_intensity_factor;   /* Raytracer.cpp:270 */
  double ambient_factor;           /* Raytracer.cpp:258 */
  double ray_offset_epsilon;       /* Raytracer.cpp:171,322,426 */
  double adaptive_epsilon_scale;
  double pure_glass_threshold;     /* Raytracer.cpp:307 */
  double exposure;
  double gamma;
  uint64_t seed;      /* seed of the addressed random function (mirt_rng.h) */
  int32_t sampling;   /* MIRT_SAMPLING_* */
  int32_t kernel_timing; /* nonzero: bracket every kernel launch with events and fill the *_kernel_ms
                            statistics (costs a few microseconds per launch); zero: device_ms only */
} mirt_config;

/* Counters and timings of one mirt_render* call. */
typedef struct mirt_stats {
  uint64_t primary_rays;     /* camera samples traced */
  uint64_t secondary_rays;   /* reflection, glossy and refraction rays traced */
  uint64_t shadow_segments;  /* closest-hit queries issued for shadows (Raytracer.cpp:181-203) */
  uint64_t node_tests;       /* ray/box tests of the primary and secondary rays, counted per ray */
  uint64_t prim_tests;       /* their ray/primitive tests (g_intersection_tests, Bvh.cpp:246) */
  uint64_t shadow_node_tests; /* ray/box tests of the shadow segments */
  uint64_t shadow_prim_tests; /* ray/primitive tests of the shadow segments */
  uint64_t launches;         /* kernel launches */
  double device_ms;          /* HIP-event time from first launch to last, on the engine's stream */
  double trace_kernel_ms;    /* sum of HIP-event durations of the trace/shade launches */
  double shadow_kernel_ms;   /* sum of HIP-event durations of the shadow launches */
  uint64_t trace_kernel_launches;
  uint64_t shadow_kernel_launches;
  uint64_t frame_attempts; /* 1, plus one for every time the frame was redone with finer launches because a queue overflowed */
} mirt_stats;

typedef struct mirt_scene mirt_scene;

/* Number of HIP devices visible to the process, or a negative mirt_status. */
int mirt_device_count(void);

/* Fills *cfg with the defaults of RenderConfig (Config.h) and jittered sampling, seed 0. */
void mirt_config_defaults(mirt_config *cfg);

/* Validates `desc`, derives the engine's device layout from it and uploads it to HIP device
 * `device`.  `desc` and everything it points to may be released after the call returns.
 * Stands for `Raytracer::Raytracer(const Scene&)` (Raytracer.cpp:26-71) minus the BVH build, which
 * the host does (it is part of `desc`). */
int mirt_scene_create(const mirt_scene_desc *desc, int device, mirt_scene **out);
void mirt_scene_destroy(mirt_scene *scene);

/* Where the BVH comes from.  The reference builds it on the host (build_bvh, Bvh.cpp:194-242) and so does
 * MIRT_BUILDER_HOST: `desc->nodes` is that tree, every walk can use it, ties and flat boxes resolve as in
 * the reference.  MIRT_BUILDER_DEVICE ignores desc->nodes / desc->leaf_prims (they may be NULL / 0, so the
 * caller need not build a tree at all) and builds one on the GPU for large scenes: Morton codes of the
 * primitives' box centres, radix sort, a binary radix tree, boxes fitted bottom-up (a moving primitive's
 * box is the union of its boxes at matrix_t0 and matrix_t1, Sphere.cpp:85-99), emitted in the 4-wide
 * form of the WIDE traversal -- the only traversal such a scene has.  The closest hit of a ray does not
 * depend on the tree except where the reference's own rules make it: a box of zero thickness is never
 * hit (BoundingBox.h:27), so a scene with planes is refused (MIRT_ERR_UNSUPPORTED), and hits at exactly
 * equal distances may be won by the other surface (Bvh.cpp:312-332): there is no reference order to restore. */
enum { MIRT_BUILDER_HOST = 0, MIRT_BUILDER_DEVICE = 1 };
typedef struct mirt_build_options {
  int32_t builder;    /* MIRT_BUILDER_* */
  int32_t leaf_size;  /* DEVICE: most primitives of a leaf, 1 .. 4; 0: default = 2, the reference's MAX_LEAF_SIZE (Bvh.h:12).
                         1 is faster on a million primitives (356 against 334 Mrays/s: fewer primitive tests), but
                         a ray then meets fewer of the far, tiny tori whose quartic the reference itself cannot
                         resolve, and frames of such scenes move from 65 to 51 dB against the reference's */
  int32_t reserved[6];
} mirt_build_options;
/* mirt_scene_create with a choice of builder; options == NULL is mirt_scene_create. */
int mirt_scene_create_with(const mirt_scene_desc *desc, int device, const mirt_build_options *options,
                           mirt_scene **out);

/* What a scene handle holds and what creating it cost (no counterpart in the reference, which logs its
 * BVH build time and node counts, Raytracer.cpp:38-56). */
typedef struct mirt_scene_info {
  int32_t builder;         /* MIRT_BUILDER_* */
  int32_t traversal;       /* MIRT_TRAVERSAL_* in use (never AUTO) */
  int32_t n_prims;
  int32_t n_wide_nodes;    /* 4-wide nodes and leaves of the WIDE traversal's tree */
  int32_t n_leaves;
  int32_t wide_depth;      /* its levels below the root */
  int32_t bvh_depth;       /* HOST: depth of desc->nodes */
  int32_t shadow_any_hit;  /* every primitive is opaque or perfectly clear: shadow rays stop at the first opaque hit
                              (PACKET and WIDE traversals; MIRT_SHADOW_ANY_HIT=0 in the environment: never) */
  double create_ms;        /* wall clock of mirt_scene_create*, uploads included */
  double tree_ms;          /* of its BVH part (HOST: skip links, 4-wide collapse, upload; DEVICE: the build) */
  double refit_ms;         /* of the last mirt_scene_refit */
} mirt_scene_info;
int mirt_scene_get_info(const mirt_scene *scene, mirt_scene_info *info);

/* Moving the primitives of a scene whose tree was built on the device (animation: the next frame's
 * transforms).  mirt_scene_update_prims replaces the records of primitives first .. first + count - 1
 * (kind, material and has_motion must stay as they are); mirt_scene_refit then fits the existing tree
 * to the new boxes, leaves up, without sorting or re-linking anything -- the tree stays correct for any
 * placement and gets slower the further primitives stray from where they were built.  The reference
 * rebuilds its tree for every frame it renders (Raytracer.cpp:26-46).  No frame may be in flight. */
int mirt_scene_update_prims(mirt_scene *scene, int first, int count, const mirt_prim *prims);
int mirt_scene_refit(mirt_scene *scene);

/* Renders one frame; stands for `Raytracer::render` (Raytracer.cpp:475-607).
 * rgb_out: caller-owned HOST buffer of height * width * 3 doubles, row-major, the post-gamma,
 * un-clamped values the reference stores in Image.pixels (Raytracer.cpp:544-570).
 * stats may be NULL. */
int mirt_render(mirt_scene *scene, const mirt_camera *camera, const mirt_config *config, int width,
                int height, int spp, double *rgb_out, mirt_stats *stats);

/* The same frame, sharded for one-process-per-GPU rendering: the image is cut into tiles of
 * tile_w x tile_h pixels, tiles_x = ceil(width / tile_w) of them per row, and this call renders the tiles
 * that belong to `rank`: tile (tx, ty) belongs to rank (ty * tiles_x + tx) % world -- except when tiles_x
 * is a multiple of world (which would hand every rank whole columns of tiles): then to rank
 * (tx + k * ty) % world with k = round(sqrt(world)), a lattice that spreads a rank's tiles evenly in both
 * directions; mirt_tile_owner() is that rule.  d_rgb_out is a DEVICE pointer
 * (on the scene's device) to
 * mirt_tile_buffer_pixels(...) * 3 doubles: the owned tiles in row-major order, each tile stored
 * row-major as tile_h rows of tile_w pixels (edge tiles are padded; padding is written as 0).
 * With world == 1 and tile_w == width, tile_h == height this is the whole frame, row-major.
 * `stream` is a hipStream_t; the call synchronises it before returning.  NULL selects the engine's own
 * (non-blocking) stream, which is first made to wait for the work already enqueued on the null stream,
 * so that a fill of d_rgb_out issued there beforehand is ordered before the frame.
 * No counterpart in the reference (it has one address space). */
int mirt_render_tiles(mirt_scene *scene, const mirt_camera *camera, const mirt_config *config,
                      int width, int height, int spp, int tile_w, int tile_h, int rank, int world,
                      double *d_rgb_out, void *stream, mirt_stats *stats);
int64_t mirt_tile_buffer_pixels(int width, int height, int tile_w, int tile_h, int rank, int world);
/* The rank that renders tile (tx, ty) of a grid tiles_x tiles wide, or -1 for arguments out of range. */
int mirt_tile_owner(int tx, int ty, int tiles_x, int world);

/* Parity hook: closest hit of the camera ray through every pixel centre at time 0.5, i.e.
 * `intersect_bvh(camera.get_ray(x+.5, y+.5, w, h, .5), root, scene, hit, 1e-5, DBL_MAX)`
 * (Bvh.cpp:306, called as in Raytracer.cpp:303).  Host buffers of width*height entries;
 * prim_id is -1 on a miss; t, normal (3 per pixel, face-forwarded unit normal) and uv (2 per pixel)
 * may each be NULL. */
int mirt_trace_ids(mirt_scene *scene, const mirt_camera *camera, int width, int height,
                   int32_t *prim_id, double *t, double *normal, double *uv);

/* Parity hook: closest hit of n caller-supplied rays, i.e.
 * `intersect_bvh(Ray(origin, dir, time), root, scene, hit, t_min, t_max)` (Bvh.cpp:306) per ray.
 * origin, dir: 3 doubles per ray (dir is used as given, not normalised); time: per ray, or NULL
 * for 0.5 (Ray.h:15); t_max: per ray, or NULL for DBL_MAX.  Outputs as in mirt_trace_ids. */
int mirt_trace_rays(mirt_scene *scene, int64_t n, const double *origin, const double *dir,
                    const double *time, double t_min, const double *t_max, int32_t *prim_id, double *t,
                    double *normal);

/* mirt_render_tiles in two halves, so that a caller can keep several frames in flight (no counterpart
 * in the reference, which renders one frame per process run).  _begin enqueues the whole frame on
 * `stream` (NULL: the scene's own) using working-memory set `slot` (0 .. MIRT_SLOTS - 1; a set's
 * queues are allocated when it is first used) and returns without
 * waiting; d_rgb_out is complete when the work enqueued on `stream` has run, so work the caller
 * enqueues on the same stream afterwards (a gather, a copy) sees the frame.  _end waits for the
 * slot's frame and fills *stats.  If a ray queue overflowed, _end renders the frame again with finer
 * launches before it returns (stats->frame_attempts > 1): whatever the caller enqueued behind _begin
 * then read an incomplete frame and has to be redone.  A slot holds one frame at a time. */
#define MIRT_SLOTS 4
int mirt_render_tiles_begin(mirt_scene *scene, const mirt_camera *camera, const mirt_config *config,
                            int width, int height, int spp, int tile_w, int tile_h, int rank, int world,
                            double *d_rgb_out, void *stream, int slot);
int mirt_render_tiles_end(mirt_scene *scene, int slot, mirt_stats *stats);

/* mirt_render_tiles for callers that cannot allocate (tight serving loops, harnesses that forbid
 * hipMalloc, code that must not disturb an allocator).  The call still waits for the frame and reads its
 * counters back, so it is not for use under HIP graph capture.  No counterpart in the reference.  Same arguments, same
 * frame, same statistics -- but the call allocates nothing, on the device or on the host: its working
 * memory (control block, pixel accumulator, shadow queue, two ray queues) is a carve-up of one static
 * array of mirt_noalloc_scratch_bytes() = 64 MiB that each GPU the library runs on holds until the
 * process ends, and its read-back buffer and events are made by mirt_scene_create.  A frame is cut
 * into as many primary launches as the queues need; a frame that does not fit even so -- the
 * accumulator (24 bytes per pixel of this rank's tiles) plus the smallest launch (one 16x16 pixel
 * block, all samples) exceed the array, or more than 255 launch-levels are needed; about a megapixel
 * at 16 spp -- is refused with MIRT_ERR_UNSUPPORTED and a message that says which limit was hit.
 * mirt_config.kernel_timing is ignored (per-launch events are created on demand).
 * All scenes on a GPU share that GPU's array: calls that overlap on one GPU (two threads, two scene
 * handles) race; calls on different GPUs do not. */
int mirt_render_tiles_noalloc(mirt_scene *scene, const mirt_camera *camera, const mirt_config *config,
                              int width, int height, int spp, int tile_w, int tile_h, int rank, int world,
                              double *d_rgb_out, void *stream, mirt_stats *stats);
uint64_t mirt_noalloc_scratch_bytes(void);

/* Progress feed.  The reference times every image row and appends "row,milliseconds" to
 * Output/render_row_times.csv as the rows finish (Raytracer.cpp:518-521,527,573-584).  The engine
 * renders all rows at once; with recording on, every block of primary samples notes the device's wall
 * clock when its samples have been shaded, per band of 16 pixel rows (cost: two atomics per 256
 * samples).  mirt_scene_progress returns the number of bands of the last frame and fills
 * band_ms[0 .. min(bands, max_bands)) with the time, in milliseconds since the frame's first block
 * started, at which the last primary sample of band b (pixel rows 16b .. 16b+15) was shaded -- shadow
 * and secondary rays of the band may still be in flight then.  Bands this rank owns no tile of read 0.
 * The feed is one buffer per scene: record with one frame in flight (frames that overlap mix their
 * marks).  mirt_render_tiles_noalloc leaves the feed out. */
int mirt_scene_record_progress(mirt_scene *scene, int enable);
int mirt_scene_progress(mirt_scene *scene, double *band_ms, int max_bands);

/* Engine tuning, no counterpart in the reference: how rays walk the BVH.
 * PACKET (one node index per wave) and LANE (one per ray) test the same boxes and primitives in the
 * same order per ray as intersect_bvh (Bvh.cpp:306-333) and return the same hits, exact ties included.
 * ORDERED (per ray, nearer child first, farther child on a per-lane stack) tests fewer boxes; it
 * returns the same closest hit except that two surfaces at exactly the same distance may swap.
 * WIDE walks the same tree collapsed into 4-wide nodes, nearer child first like ORDERED, and runs the primitive
 * tests a kind at a time across the wave: the walk for large scenes.  Same closest hit; a closest-hit ray that
 * meets two surfaces within a relative 1e-9 of one distance (MIRT_TIE_BAND in the environment; 0: off) is walked
 * again as LANE walks, so coincident surfaces resolve as in the reference whenever the scene carries the
 * reference's tree (MIRT_BUILDER_HOST); a tree built on the device has no reference order to restore.
 * AUTO: PACKET up to 512 primitives, WIDE above. */
enum { MIRT_TRAVERSAL_AUTO = 0, MIRT_TRAVERSAL_PACKET = 1, MIRT_TRAVERSAL_LANE = 2, MIRT_TRAVERSAL_ORDERED = 3,
       MIRT_TRAVERSAL_WIDE = 4 };
int mirt_scene_set_traversal(mirt_scene *scene, int mode);

/* Several GPUs in one process.  No counterpart in the reference, whose only parallelism is the OpenMP
 * row loop (Raytracer.cpp:525): a group replicates the scene on every device named (one mirt_scene
 * each), mirt_group_render deals the image's 16x16 tiles round-robin (mirt_tile_owner), lets
 * all members render their share at the same time, gathers the shares on the first device with RCCL
 * (ncclSend / ncclRecv over xGMI; hipMemcpyPeerAsync where RCCL is missing or a device is named twice
 * -- mirt_group_transport() names what is in use) and writes the frame to rgb_out (host, as
 * mirt_render).  stats: sums over the members, device_ms their maximum; member_stats: NULL or
 * mirt_group_size() entries, one per member. */
#define MIRT_GROUP_MAX 16
#define MIRT_GROUP_TILE 16
typedef struct mirt_group mirt_group;
int mirt_group_create(const mirt_scene_desc *desc, const int *devices, int n_devices, mirt_group **out);
/* The same with build options for every member's replica (mirt_scene_create_with): with MIRT_BUILDER_DEVICE each
 * GPU builds its own tree from the primitives, and `desc` need not carry one. */
int mirt_group_create_with(const mirt_scene_desc *desc, const int *devices, int n_devices,
                           const mirt_build_options *options, mirt_group **out);
void mirt_group_destroy(mirt_group *group);
int mirt_group_size(const mirt_group *group);
const char *mirt_group_transport(const mirt_group *group);
int mirt_group_set_traversal(mirt_group *group, int mode);
int mirt_group_render(mirt_group *group, const mirt_camera *camera, const mirt_config *config, int width,
                      int height, int spp, double *rgb_out, mirt_stats *stats, mirt_stats *member_stats);
/* mirt_group_render in two halves, for frames that follow one another (no counterpart in the reference).
 * _begin enqueues the whole frame -- every member's share on that member's device, the gather of the
 * shares behind them, the kernel that puts the tiles in place on the first device -- and returns without
 * waiting for any device; `slot` (0 .. MIRT_GROUP_SLOTS - 1) names the set of buffers and streams it uses,
 * so a second frame can be begun while the first is still running.  _end waits for the slot's frame, copies
 * it to rgb_out when that is not NULL (the read-back runs on a stream of its own: begin frame i + 1 first
 * and it overlaps that frame's rendering), and fills the statistics as mirt_group_render does.  After _end,
 * and until the slot's next _begin, mirt_group_frame(group, slot) is the frame in the first device's
 * memory (height * width * 3 doubles, row-major) for callers that go on working there.  When a call
 * fails, every member has been waited for and the slot is free again. */
#define MIRT_GROUP_SLOTS 2
int mirt_group_render_begin(mirt_group *group, const mirt_camera *camera, const mirt_config *config, int width,
                            int height, int spp, int slot);
int mirt_group_render_end(mirt_group *group, int slot, double *rgb_out, mirt_stats *stats, mirt_stats *member_stats);
const double *mirt_group_frame(const mirt_group *group, int slot);

const char *mirt_last_error(void);
int mirt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
