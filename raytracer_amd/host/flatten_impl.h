// Scene -> flat arrays, as templates over the scene classes.
//
// The drop-in classes (mirt_scene.h) and the reference's own (Code/Core/Scene.h:31-41, Code/Geometry/*.h,
// Code/Core/{Material,Light,Camera,Config}.h, Code/IO/Image.h, Code/Math/Transform.h) have the same
// members under the same names, so one text serves both: flatten.cpp instantiates it for the drop-in
// classes; a maintainer of the reference includes this header next to the reference's headers and calls
// mirt_flatten::scene<Image>(scene) on the reference's Scene (integration/reference_binding.cpp does, and
// tests/test_reference_binding.py compares the two results byte for byte).  Nothing here names a class:
// scene, shape, box, transform and image types are template parameters or deduced from members.
//
// BVH: same tree as the reference builds, because a different topology can change which objects
// are visible (a zero-thickness box is never hit by BoundingBox::intersect, BoundingBox.h:27) and
// which object wins an exact tie (leaf order, Bvh.cpp:314-321).  Rule (Bvh.cpp:194-242):
//   box  = union of the objects' world boxes (get_*_bounding_box, e.g. Sphere.cpp:82-103)
//   leaf = at most MAX_LEAF_SIZE (2) objects, or depth >= MAX_DEPTH (30)
//   axis = x only if its extent is strictly the largest, else y if larger than z, else z (Bvh.cpp:89-94)
//   cut  = std::nth_element at n/2 on the objects' static centres (Bvh.cpp:164-173, :108-145)
// The reference copies index vectors at every level; here one index array is partitioned in place,
// which visits the same elements in the same order, and its final state is the leaf list.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>

#include "flat_scene.h"

namespace mirt_flatten {

constexpr int kMaxLeafSize = 2;
constexpr int kMaxDepth = 30;

struct Box {  // world box of a primitive or a node
  double lo[3], hi[3];
  void grow(const Box &o) {
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::min(lo[a], o.lo[a]);
      hi[a] = std::max(hi[a], o.hi[a]);
    }
  }
};
template <class BB> Box box_of(const BB &b) { return Box{{b.min.x, b.min.y, b.min.z}, {b.max.x, b.max.y, b.max.z}}; }

// get_*_bounding_box (e.g. Sphere.cpp:82-103): the object box through cached_transform, or, for a moving
// shape, the union of its boxes through matrix_t0 and matrix_t1
template <class ShapeT> Box world_box(const ShapeT &s, const double (&lo)[3], const double (&hi)[3]) {
  using TransformT = std::decay_t<decltype(s.cached_transform)>;
  using BB = std::decay_t<decltype(s.cached_transform.transform_bbox({}))>;
  using PointT = std::decay_t<decltype(BB::min)>;
  const BB object_box{PointT(lo[0], lo[1], lo[2]), PointT(hi[0], hi[1], hi[2])};
  if (!s.has_motion) return box_of(s.cached_transform.transform_bbox(object_box));
  const TransformT t0(s.start_transform), t1(s.end_transform);
  Box b = box_of(t0.transform_bbox(object_box));
  b.grow(box_of(t1.transform_bbox(object_box)));
  return b;
}

struct Builder {
  std::vector<Box> boxes;               // per primitive, world space
  std::vector<double> centres;          // per primitive, split key: 3 doubles
  std::vector<int32_t> order;           // primitive indices, partitioned in place
  FlatScene *out = nullptr;
  // nodes in the subtree over n objects at depth d: the split rule depends on n and d only, so every
  // node's index in depth-first order is known before its subtree is built
  std::map<std::pair<int, int>, int32_t> sizes;

  int32_t subtree_nodes(int n, int depth) {
    if (n <= kMaxLeafSize || depth >= kMaxDepth) return 1;
    const auto key = std::make_pair(n, depth);
    const auto it = sizes.find(key);
    if (it != sizes.end()) return it->second;
    const int mid = n / 2;
    const int32_t v = 1 + subtree_nodes(mid, depth + 1) + subtree_nodes(n - mid, depth + 1);
    sizes[key] = v;
    return v;
  }

  // Builds the subtree of order[lo, hi) into nodes[me ...] (already allocated); returns its depth.
  // The two halves of a split touch disjoint ranges of `order` and of `nodes`: on large scenes the top
  // `fork` levels build them on two threads each.
  int build(int lo, int hi, int depth, int32_t me, int fork) {
    Box box{{0, 0, 0}, {0, 0, 0}};
    for (int i = lo; i < hi; ++i) {
      if (i == lo) box = boxes[order[i]];
      else box.grow(boxes[order[i]]);
    }
    auto fill = [&](int32_t left, int32_t right, int32_t first, int32_t count) {
      mirt_node &n = out->nodes[me];
      for (int a = 0; a < 3; ++a) {
        n.bmin[a] = box.lo[a];
        n.bmax[a] = box.hi[a];
      }
      n.left = left; n.right = right; n.first = first; n.count = count;
    };
    const int n = hi - lo;
    if (n <= kMaxLeafSize || depth >= kMaxDepth) {
      fill(-1, -1, lo, n);
      return 1;
    }
    const double ex = box.hi[0] - box.lo[0], ey = box.hi[1] - box.lo[1], ez = box.hi[2] - box.lo[2];
    const int axis = (ex > ey && ex > ez) ? 0 : (ey > ez ? 1 : 2);
    const int mid = n / 2;  // n >= 3, so both halves are non-empty
    std::nth_element(order.begin() + lo, order.begin() + lo + mid, order.begin() + hi,
                     [this, axis](int32_t a, int32_t b) { return centres[3 * a + axis] < centres[3 * b + axis]; });
    const int32_t left = me + 1, right = left + sizes_at(mid, depth + 1);
    int dl = 0, dr = 0;
    if (fork > 0) {
      std::thread t([&] { dl = build(lo, lo + mid, depth + 1, left, fork - 1); });
      dr = build(lo + mid, hi, depth + 1, right, fork - 1);
      t.join();
    } else {
      dl = build(lo, lo + mid, depth + 1, left, 0);
      dr = build(lo + mid, hi, depth + 1, right, 0);
    }
    fill(left, right, 0, 0);
    return 1 + std::max(dl, dr);
  }
  int32_t sizes_at(int n, int depth) const {  // read-only once run() has filled the table
    if (n <= kMaxLeafSize || depth >= kMaxDepth) return 1;
    return sizes.at(std::make_pair(n, depth));
  }

  void run() {
    const int total = static_cast<int>(order.size());
    const int32_t n_nodes = subtree_nodes(total, 0);
    out->nodes.assign(static_cast<std::size_t>(n_nodes), mirt_node{});
    int fork = total >= 65536 ? 4 : 0;  // 16 threads at the fourth level
    if (const char *e = std::getenv("MIRT_BVH_FORK_LEVELS")) fork = std::max(0, std::min(6, std::atoi(e)));
    out->max_depth = build(0, total, 0, 0, fork);
    out->node_count = n_nodes;
    out->leaf_count = 0;
    for (const mirt_node &n : out->nodes) out->leaf_count += n.left < 0 ? 1 : 0;
  }
};

template <class V> void copy3(double *dst, const V &v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
template <class M> void copy16(double *dst, const M &m) { std::memcpy(dst, m.m, sizeof(double) * 16); }

template <class ShapeT> mirt_prim base_prim(const ShapeT &s, int type, int material) {
  mirt_prim p;
  std::memset(&p, 0, sizeof p);
  p.type = type;
  p.visible = s.visible ? 1 : 0;
  p.has_motion = s.has_motion ? 1 : 0;
  p.material = material;
  copy16(p.o2w, s.cached_transform.object_to_world);
  copy16(p.w2o, s.cached_transform.world_to_object);
  copy16(p.m0, s.start_transform);
  copy16(p.m1, s.end_transform);
  return p;
}

// Name -> loaded image registry standing for the reference's TextureManager (Texture.cpp:11-71).
template <class ImageT> struct TextureRegistry {
  FlatScene *out;
  std::string root;
  std::map<std::string, int> index;

  void load(const std::string &name) {
    if (name.empty() || index.count(name)) return;
    std::string file = name;
    const auto jpg = file.find(".jpg");  // only PPM can be read: tex.jpg -> tex.ppm
    if (jpg != std::string::npos) file.replace(jpg, 4, ".ppm");
    std::string path = root + "Textures/" + file;
    if (!std::ifstream(path).good()) path = root + "../Textures/" + file;
    ImageT img(path);
    if (img.width == 0 || img.height == 0) {
      std::cerr << "[ERROR] Failed to load texture (invalid dimensions) file=" << name << "\n";
      return;
    }
    std::vector<uint16_t> px(static_cast<std::size_t>(img.width) * img.height * 3);
    std::size_t k = 0;
    for (const auto &row : img.pixels)
      for (const auto &p : row)
        for (int c = 0; c < 3; ++c) px[k++] = static_cast<uint16_t>(std::lround(p[c] * img.max_value));
    index[name] = static_cast<int>(out->textures.size());
    mirt_texture t;
    std::memset(&t, 0, sizeof t);
    t.width = img.width;
    t.height = img.height;
    t.max_value = img.max_value;
    out->texels.push_back(std::move(px));
    out->textures.push_back(t);
    out->texture_names.push_back(name);
  }
  int find(const std::string &name) const {
    if (name.empty()) return -1;
    const auto it = index.find(name);
    return it == index.end() ? -1 : it->second;
  }
};

template <class MaterialT, class Registry> mirt_material material(const MaterialT &m, const Registry &reg) {
  mirt_material o;
  std::memset(&o, 0, sizeof o);
  copy3(o.diffuse, m.diffuse_color);
  copy3(o.specular, m.specular_color);
  copy3(o.ambient, m.ambient_color);
  copy3(o.emission, m.emission_color);
  o.shininess = m.shininess;
  o.glossiness = m.glossiness;
  o.reflectivity = m.reflectivity;
  o.transparency = m.transparency;
  o.refractive_index = m.refractive_index;
  o.emission_strength = m.emission_strength;
  o.bump_strength = m.bump_strength;
  o.has_texture = m.has_texture ? 1 : 0;
  o.texture = reg.find(m.texture_file);
  o.normal_map = reg.find(m.normal_map);
  o.bump_map = reg.find(m.bump_map);
  return o;
}

// The whole scene.  ImageT: the class set's Image (PPM reader: ImageT(path), .width, .height, .max_value,
// .pixels[y][x][c]).  texture_root: directory that holds `Textures/`; empty = the current directory, as
// in the reference (Code/IO/Texture.cpp:37-45).  build_tree = false leaves nodes and leaf_prims empty (for
// a tree built on the device).
template <class ImageT, class SceneT>
FlatScene scene(const SceneT &scene, const std::string &texture_root = "", bool build_tree = true) {
  FlatScene fs;
  std::string root = texture_root;
  if (!root.empty() && root.back() != '/') root.push_back('/');

  // textures first: only the maps named by spheres, cubes and planes are ever loaded
  // (Raytracer.cpp:94-108); other shapes can use an image only if one of those loaded it
  TextureRegistry<ImageT> reg{&fs, root, {}};
  auto preload = [&reg](const auto &m) {
    if (m.has_texture) reg.load(m.texture_file);
    reg.load(m.normal_map);
    reg.load(m.bump_map);
  };
  for (const auto &s : scene.spheres) preload(s.material);
  for (const auto &s : scene.cubes) preload(s.material);
  for (const auto &s : scene.planes) preload(s.material);
  for (std::size_t i = 0; i < fs.textures.size(); ++i) fs.textures[i].rgb = fs.texels[i].data();

  Builder b;
  b.out = &fs;
  // Global primitive numbering (Bvh.cpp:244-304): spheres, cubes, planes, toruses, cylinders, cones.  Every
  // primitive has its own material record, so a primitive's index is also its material's; the records of a
  // kind are independent of one another and large kinds are filled on several threads.
  const std::size_t n_kind[6] = {scene.spheres.size(), scene.cubes.size(),     scene.planes.size(),
                                 scene.toruses.size(), scene.cylinders.size(), scene.cones.size()};
  std::size_t first[6], total = 0;
  for (int k = 0; k < 6; ++k) {
    first[k] = total;
    total += n_kind[k];
  }
  fs.prims.resize(total);
  fs.materials.resize(total);
  b.boxes.resize(total);
  b.centres.resize(3 * total);
  auto put = [&](std::size_t at, const auto &s, const mirt_prim &p, const Box &box, const auto &centre) {
    fs.materials[at] = material(s.material, reg);
    fs.prims[at] = p;
    b.boxes[at] = box;
    b.centres[3 * at] = centre.x;
    b.centres[3 * at + 1] = centre.y;
    b.centres[3 * at + 2] = centre.z;
  };
  auto for_each_index = [](std::size_t n, auto &&body) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const std::size_t n_threads = n >= 16384 ? std::min<std::size_t>(std::min(hw, 16u), n / 8192) : 1;
    if (n_threads <= 1) {
      for (std::size_t i = 0; i < n; ++i) body(i);
      return;
    }
    std::vector<std::thread> pool;
    for (std::size_t t = 0; t < n_threads; ++t)
      pool.emplace_back([&body, n, t, n_threads]() {
        for (std::size_t i = n * t / n_threads; i < n * (t + 1) / n_threads; ++i) body(i);
      });
    for (auto &th : pool) th.join();
  };
  const double unit_lo[3] = {-1, -1, -1}, unit_hi[3] = {1, 1, 1};

  for_each_index(n_kind[0], [&](std::size_t i) {
    const auto &s = scene.spheres[i];
    const std::size_t at = first[0] + i;
    put(at, s, base_prim(s, MIRT_SPHERE, static_cast<int>(at)), world_box(s, unit_lo, unit_hi), s.location);
  });
  for_each_index(n_kind[1], [&](std::size_t i) {
    const auto &s = scene.cubes[i];
    const std::size_t at = first[1] + i;
    mirt_prim p = base_prim(s, MIRT_CUBE, static_cast<int>(at));
    p.param[0] = s.scale.x; p.param[1] = s.scale.y; p.param[2] = s.scale.z;  // box UVs (Cube.cpp:106-131)
    put(at, s, p, world_box(s, unit_lo, unit_hi), s.translation);
  });
  for (std::size_t i = 0; i < n_kind[2]; ++i) {  // (planes append their corner points: one after the other)
    const auto &s = scene.planes[i];
    const std::size_t at = first[2] + i;
    mirt_prim p = base_prim(s, MIRT_PLANE, static_cast<int>(at));
    p.first_point = static_cast<int32_t>(fs.plane_points.size() / 3);
    p.n_points = static_cast<int32_t>(s.points.size());
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    using PointT = typename std::decay_t<decltype(s.points)>::value_type;
    PointT centre(0, 0, 0);
    for (std::size_t j = 0; j < s.points.size(); ++j) {
      const auto &q = s.points[j];
      fs.plane_points.insert(fs.plane_points.end(), {q.x, q.y, q.z});
      const double v[3] = {q.x, q.y, q.z};
      for (int a = 0; a < 3; ++a) {
        lo[a] = j == 0 ? v[a] : std::min(lo[a], v[a]);
        hi[a] = j == 0 ? v[a] : std::max(hi[a], v[a]);
      }
      centre = centre + q;
    }
    if (!s.points.empty()) centre = centre * (1.0 / s.points.size());
    // an empty plane keeps the degenerate box at the origin un-transformed (Plane.cpp:136-138)
    put(at, s, p, s.points.empty() ? Box{{0, 0, 0}, {0, 0, 0}} : world_box(s, lo, hi), centre);
  }
  for_each_index(n_kind[3], [&](std::size_t i) {
    const auto &s = scene.toruses[i];
    const std::size_t at = first[3] + i;
    mirt_prim p = base_prim(s, MIRT_TORUS, static_cast<int>(at));
    p.param[0] = s.major_radius; p.param[1] = s.minor_radius;
    const double R = s.major_radius + s.minor_radius, r = s.minor_radius;
    const double lo[3] = {-R, -R, -r}, hi[3] = {R, R, r};
    put(at, s, p, world_box(s, lo, hi), s.location);
  });
  for_each_index(n_kind[4], [&](std::size_t i) {
    const auto &s = scene.cylinders[i];
    const std::size_t at = first[4] + i;
    mirt_prim p = base_prim(s, MIRT_CYLINDER, static_cast<int>(at));
    p.param[0] = s.radius; p.param[1] = s.depth;
    const double r = s.radius, h = s.depth / 2.0;
    const double lo[3] = {-r, -r, -h}, hi[3] = {r, r, h};
    put(at, s, p, world_box(s, lo, hi), s.location);
  });
  for_each_index(n_kind[5], [&](std::size_t i) {
    const auto &s = scene.cones[i];
    const std::size_t at = first[5] + i;
    mirt_prim p = base_prim(s, MIRT_CONE, static_cast<int>(at));
    p.param[0] = s.radius; p.param[1] = s.depth;
    const double r = s.radius, h = s.depth / 2.0;
    const double lo[3] = {-r, -r, -h}, hi[3] = {r, r, h};
    put(at, s, p, world_box(s, lo, hi), s.location);
  });

  if (build_tree) {
    b.order.resize(total);
    for (std::size_t i = 0; i < total; ++i) b.order[i] = static_cast<int32_t>(i);
    b.run();  // an empty scene still gets its one empty leaf
    fs.leaf_prims = std::move(b.order);
  }

  for (const auto &l : scene.lights) {
    mirt_light o;
    std::memset(&o, 0, sizeof o);
    copy3(o.location, l.location);
    copy3(o.color, l.color);
    copy3(o.normal, l.normal);
    o.intensity = l.intensity;
    o.area_size[0] = l.area_size_x;
    o.area_size[1] = l.area_size_y;
    o.type = l.light_type == "AREA" ? MIRT_LIGHT_AREA : MIRT_LIGHT_POINT;
    o.area_shape = (l.area_shape == "SQUARE" || l.area_shape == "RECTANGLE") ? MIRT_AREA_RECT : MIRT_AREA_DISK;
    o.samples = l.samples;
    fs.lights.push_back(o);
  }
  copy3(fs.background_color, scene.settings.background_color);
  fs.background_strength = scene.settings.background_strength;
  return fs;
}

// Camera (Camera.h:10-30): the fields Camera::get_ray reads.
template <class CameraT> mirt_camera camera(const CameraT &c) {
  mirt_camera o;
  std::memset(&o, 0, sizeof o);
  copy3(o.location, c.location);
  copy3(o.gaze, c.gaze_direction);
  copy3(o.up, c.up_direction);
  o.focal_length = c.focal_length;
  o.sensor_height = c.sensor_height;
  o.focus_distance = c.focus_distance;
  o.aperture_fstop = c.aperture_fstop;
  o.dof_enabled = c.dof_enabled ? 1 : 0;
  return o;
}

// The engine's seed: RenderConfig::seed where the class has one (the drop-in's --seed), else 0.
template <class C, class = void> struct seed_of {
  static uint64_t get(const C &) { return 0; }
};
template <class C> struct seed_of<C, std::void_t<decltype(std::declval<const C &>().seed)>> {
  static uint64_t get(const C &c) { return static_cast<uint64_t>(c.seed); }
};

// Snapshot of the RenderConfig fields the render path reads (Config.h:11-168).
template <class ConfigT> mirt_config config(const ConfigT &c) {
  mirt_config o;
  std::memset(&o, 0, sizeof o);
  o.max_ray_depth = c.max_ray_depth;
  o.shadow_samples = c.shadow_samples;
  o.glossy_samples = c.glossy_samples;
  o.enable_motion_blur = c.enable_motion_blur ? 1 : 0;
  o.enable_textures = c.enable_textures ? 1 : 0;
  o.use_adaptive_epsilon = c.use_adaptive_epsilon ? 1 : 0;
  o.tone_mapping_mode = c.tone_mapping_mode == "reinhard" ? MIRT_TONEMAP_REINHARD
                        : c.tone_mapping_mode == "exposure" ? MIRT_TONEMAP_EXPOSURE
                                                             : MIRT_TONEMAP_NONE;
  o.enable_gamma_correction = c.enable_gamma_correction ? 1 : 0;
  o.light_intensity_factor = c.light_intensity_factor;
  o.ambient_factor = c.ambient_factor;
  o.ray_offset_epsilon = c.ray_offset_epsilon;
  o.adaptive_epsilon_scale = c.adaptive_epsilon_scale;
  o.pure_glass_threshold = c.pure_glass_threshold;
  o.exposure = c.exposure;
  o.gamma = c.gamma;
  o.seed = seed_of<ConfigT>::get(c);
  o.sampling = MIRT_SAMPLING_JITTER;
  return o;
}

}  // namespace mirt_flatten
