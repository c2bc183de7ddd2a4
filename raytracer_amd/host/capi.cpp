// C entry points of the host library (libmirt_host.so) for callers that are not C++ (the Python
// tests and bench drive the loader and the flattening through these with ctypes).  They wrap
// load_scene + flatten_scene and the PPM writer; none of them touches the GPU.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <string>

#include "flatten.h"
#include "mirt_scene.h"
#include "synthetic.h"

namespace {
struct HostScene {
  Scene scene;
  FlatScene flat;
};

void p3(std::FILE *f, const char *k, const Vec3 &v) { std::fprintf(f, "%s %.17g %.17g %.17g\n", k, v.x, v.y, v.z); }
void p1(std::FILE *f, const char *k, double v) { std::fprintf(f, "%s %.17g\n", k, v); }
void pi(std::FILE *f, const char *k, long v) { std::fprintf(f, "%s %ld\n", k, v); }
void ps(std::FILE *f, const char *k, const std::string &v) { std::fprintf(f, "%s [%s]\n", k, v.c_str()); }
void pm(std::FILE *f, const char *k, const Mat4 &m) {
  std::fprintf(f, "%s", k);
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) std::fprintf(f, " %.17g", m.m[r][c]);
  std::fprintf(f, "\n");
}
void dump_shape(std::FILE *f, const Shape &s) {
  ps(f, "  name", s.name);
  pi(f, "  visible", s.visible);
  pi(f, "  has_motion", s.has_motion);
  pm(f, "  start", s.start_transform);
  pm(f, "  end", s.end_transform);
  pm(f, "  o2w", s.cached_transform.object_to_world);
  pm(f, "  w2o", s.cached_transform.world_to_object);
  const Material &m = s.material;
  p3(f, "  diffuse", m.diffuse_color);
  p3(f, "  specular", m.specular_color);
  p3(f, "  ambient", m.ambient_color);
  p1(f, "  shininess", m.shininess);
  p1(f, "  glossiness", m.glossiness);
  p1(f, "  reflectivity", m.reflectivity);
  p1(f, "  transparency", m.transparency);
  p1(f, "  refractive_index", m.refractive_index);
  ps(f, "  texture_file", m.texture_file);
  pi(f, "  has_texture", m.has_texture);
  p3(f, "  emission", m.emission_color);
  p1(f, "  emission_strength", m.emission_strength);
  ps(f, "  normal_map", m.normal_map);
  ps(f, "  bump_map", m.bump_map);
  p1(f, "  bump_strength", m.bump_strength);
}
template <class S> void dump_placed(std::FILE *f, const S &s, const Vec3 &loc) {
  dump_shape(f, s);
  p3(f, "  location", loc);
  p3(f, "  rotation", s.rotation);
  p3(f, "  scale", s.scale);
}
}  // namespace

namespace {
thread_local std::string g_host_error;

// Runs `fill` on a fresh HostScene; no C++ exception leaves the library (the loader throws
// std::runtime_error on a truncated matrix block, as the reference does: SceneLoader.cpp:400).
template <class F> void *make_scene(F fill) {
  HostScene *h = nullptr;
  try {
    h = new HostScene;
    fill(*h);
    g_host_error.clear();
    return h;
  } catch (const std::exception &e) {
    g_host_error = e.what();
  } catch (...) {
    g_host_error = "unknown C++ exception";
  }
  delete h;
  return nullptr;
}
}  // namespace

extern "C" {

// Message of the last mirt_host_scene_load / mirt_host_scene_synthetic that returned NULL on this thread.
const char *mirt_host_last_error(void) { return g_host_error.c_str(); }

// Parses `path` and flattens it.  texture_root: directory holding Textures/ ("" = CWD).
// A missing file yields an empty scene, as load_scene does; NULL means the parser threw or memory
// ran out (mirt_host_last_error() says which).
void *mirt_host_scene_load(const char *path, const char *texture_root) {
  return make_scene([&](HostScene &h) {
    h.scene = load_scene(path ? path : "");
    h.flat = flatten_scene(h.scene, texture_root ? texture_root : "");
  });
}

// The synthetic cloud of make_synthetic_scene(n_prims, seed), flattened.
void *mirt_host_scene_synthetic(int n_prims, uint64_t seed) {
  return make_scene([&](HostScene &h) {
    h.scene = make_synthetic_scene(n_prims, seed);
    h.flat = flatten_scene(h.scene, "");
  });
}

// The same without the host BVH build (for MIRT_BUILDER_DEVICE); *flatten_ms gets the wall clock of the
// flattening alone (build_tree != 0: tree included), the scene generation not counted.
void *mirt_host_scene_synthetic_opt(int n_prims, uint64_t seed, int build_tree, double *flatten_ms) {
  return make_scene([&](HostScene &h) {
    h.scene = make_synthetic_scene(n_prims, seed);
    const auto t0 = std::chrono::steady_clock::now();
    h.flat = flatten_scene(h.scene, "", build_tree != 0);
    if (flatten_ms) *flatten_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  });
}

// Parses `path` and flattens it without building the tree.
void *mirt_host_scene_load_opt(const char *path, const char *texture_root, int build_tree) {
  return make_scene([&](HostScene &h) {
    h.scene = load_scene(path ? path : "");
    h.flat = flatten_scene(h.scene, texture_root ? texture_root : "", build_tree != 0);
  });
}

// Writes the handle's scene in the ASCII grammar (sphere / torus / cylinder / cone scenes only).
int mirt_host_scene_write_ascii(void *handle, const char *path) {
  return write_scene_ascii(static_cast<HostScene *>(handle)->scene, path ? path : "") ? 0 : -1;
}

void mirt_host_scene_free(void *handle) { delete static_cast<HostScene *>(handle); }

// The pointers in *out stay valid until mirt_host_scene_free(handle).
void mirt_host_scene_desc(void *handle, mirt_scene_desc *out) { *out = static_cast<HostScene *>(handle)->flat.desc(); }

int mirt_host_scene_camera_count(void *handle) {
  return static_cast<int>(static_cast<HostScene *>(handle)->scene.cameras.size());
}

// Camera `index` with the command-line depth-of-field override applied (Main.cpp:160-176):
// dof_mode 0 = as in the file, 1 = enable with (fstop, focus), 2 = disable.
int mirt_host_scene_camera(void *handle, int index, int dof_mode, double fstop, double focus, mirt_camera *out,
                           int *res_x, int *res_y) {
  const Scene &sc = static_cast<HostScene *>(handle)->scene;
  if (index < 0 || index >= static_cast<int>(sc.cameras.size())) return -1;
  Camera c = sc.cameras[index];
  if (dof_mode == 1 && fstop > 0.0) {
    c.dof_enabled = true;
    c.aperture_fstop = fstop;
    c.focus_distance = focus;
  } else if (dof_mode == 1 || dof_mode == 2) {
    c.dof_enabled = false;
  }
  *out = flatten_camera(c);
  if (res_x) *res_x = c.film_resolution_x;
  if (res_y) *res_y = c.film_resolution_y;
  return 0;
}

void mirt_host_bvh_stats(void *handle, int *nodes, int *leaves, int *depth) {
  const FlatScene &f = static_cast<HostScene *>(handle)->flat;
  *nodes = f.node_count;
  *leaves = f.leaf_count;
  *depth = f.max_depth;
}

// Canonical text dump of the parsed Scene; the golden harness prints the reference's Scene in the
// same format, so the two files must be identical byte for byte.
int mirt_host_scene_dump(void *handle, const char *out_path) {
  const Scene &sc = static_cast<HostScene *>(handle)->scene;
  std::FILE *f = std::fopen(out_path, "w");
  if (!f) return -1;
  p3(f, "background_color", sc.settings.background_color);
  p1(f, "background_strength", sc.settings.background_strength);
  pi(f, "cameras", sc.cameras.size());
  for (const auto &c : sc.cameras) {
    ps(f, "camera", c.name);
    p3(f, "  location", c.location);
    p3(f, "  gaze", c.gaze_direction);
    p3(f, "  up", c.up_direction);
    p1(f, "  focal", c.focal_length);
    p1(f, "  sensor_w", c.sensor_width);
    p1(f, "  sensor_h", c.sensor_height);
    pi(f, "  res_x", c.film_resolution_x);
    pi(f, "  res_y", c.film_resolution_y);
    pi(f, "  dof_enabled", c.dof_enabled);
    p1(f, "  focus_distance", c.focus_distance);
    p1(f, "  aperture_fstop", c.aperture_fstop);
  }
  pi(f, "lights", sc.lights.size());
  for (const auto &l : sc.lights) {
    ps(f, "light", l.name);
    p3(f, "  location", l.location);
    p1(f, "  intensity", l.intensity);
    p3(f, "  color", l.color);
    ps(f, "  light_type", l.light_type);
    ps(f, "  area_shape", l.area_shape);
    p1(f, "  area_size_x", l.area_size_x);
    p1(f, "  area_size_y", l.area_size_y);
    pi(f, "  samples", l.samples);
    p3(f, "  normal", l.normal);
  }
  pi(f, "spheres", sc.spheres.size());
  for (const auto &s : sc.spheres) dump_placed(f, s, s.location);
  pi(f, "cubes", sc.cubes.size());
  for (const auto &s : sc.cubes) dump_placed(f, s, s.translation);
  pi(f, "planes", sc.planes.size());
  for (const auto &s : sc.planes) {
    dump_shape(f, s);
    pi(f, "  points", s.points.size());
    for (const auto &q : s.points) p3(f, "   pt", q);
  }
  pi(f, "toruses", sc.toruses.size());
  for (const auto &s : sc.toruses) {
    dump_placed(f, s, s.location);
    p1(f, "  major_radius", s.major_radius);
    p1(f, "  minor_radius", s.minor_radius);
  }
  pi(f, "cylinders", sc.cylinders.size());
  for (const auto &s : sc.cylinders) {
    dump_placed(f, s, s.location);
    p1(f, "  radius", s.radius);
    p1(f, "  depth", s.depth);
  }
  pi(f, "cones", sc.cones.size());
  for (const auto &s : sc.cones) {
    dump_placed(f, s, s.location);
    p1(f, "  radius", s.radius);
    p1(f, "  depth", s.depth);
  }
  std::fclose(f);
  return 0;
}

// Depth-first text dump of the flattened BVH in the format of the golden harness's `dump-bvh`.
int mirt_host_bvh_dump(void *handle, const char *out_path) {
  const FlatScene &fs = static_cast<HostScene *>(handle)->flat;
  std::FILE *f = std::fopen(out_path, "w");
  if (!f) return -1;
  struct Walk {
    const FlatScene &fs;
    std::FILE *f;
    void go(int i, int depth) const {
      const mirt_node &n = fs.nodes[i];
      const bool leaf = n.left < 0;
      std::fprintf(f, "node depth=%d leaf=%d min %.17g %.17g %.17g max %.17g %.17g %.17g idx", depth, (int)leaf,
                   n.bmin[0], n.bmin[1], n.bmin[2], n.bmax[0], n.bmax[1], n.bmax[2]);
      if (leaf)
        for (int k = 0; k < n.count; ++k) std::fprintf(f, " %d", fs.leaf_prims[n.first + k]);
      std::fprintf(f, "\n");
      if (!leaf) {
        go(n.left, depth + 1);
        go(n.right, depth + 1);
      }
    }
  };
  if (!fs.nodes.empty()) Walk{fs, f}.go(0, 0);
  std::fclose(f);
  return 0;
}

// Writes post-gamma doubles (row-major, 3 per pixel) as the reference writes its frame: ASCII P3,
// clamp + lround(v * 255) (Image.cpp:123-128,184-204).  binary != 0 selects P6.
int mirt_host_write_ppm(const char *path, const double *rgb, int width, int height, int binary) {
  Image img(height, width, 255, binary ? "P6" : "P3");
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const double *p = rgb + (static_cast<std::size_t>(y) * width + x) * 3;
      img.pixels[y][x] = Pixel(p[0], p[1], p[2]);
    }
  img.write(path);
  return 0;
}

// Reads a P3/P6 file with the reference's reader quirks; rgb (may be NULL to query the size) gets
// width*height*3 doubles in [0,1].
int mirt_host_read_ppm(const char *path, int *width, int *height, int *max_value, double *rgb) {
  Image img{std::string(path)};
  *width = img.width;
  *height = img.height;
  *max_value = img.max_value;
  if (img.width == 0 || img.height == 0) return -1;
  if (rgb)
    for (int y = 0; y < img.height; ++y)
      for (int x = 0; x < img.width; ++x) {
        double *p = rgb + (static_cast<std::size_t>(y) * img.width + x) * 3;
        p[0] = img.pixels[y][x].x;
        p[1] = img.pixels[y][x].y;
        p[2] = img.pixels[y][x].z;
      }
  return 0;
}

void mirt_host_config_defaults(mirt_config *out) { *out = flatten_config(RenderConfig()); }

}  // extern "C"
