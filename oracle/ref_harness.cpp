// TEST INFRASTRUCTURE ONLY -- never linked into or called by product code.
//
// Golden-vector harness around the UNMODIFIED reference renderer.  It is compiled against
// the reference's own headers and linked with the reference's own objects (everything but
// Code/Core/Main.cpp), built by oracle/Makefile from the sources where they lie under
// /root/reference into oracle/_ref/.  Nothing of the reference is copied here: this file
// only *calls* it -- load_scene (SceneLoader.cpp:112), Raytracer::Raytracer
// (Raytracer.cpp:26), intersect_bvh (Bvh.cpp:306), Raytracer::trace (Raytracer.cpp:296),
// Camera::get_ray (Camera.cpp:10) -- and writes what it returns.
//
// Randomness: the two generators of the reference are fed by ref_rng_hook.h from the
// addressed stream of include/mirt_rng.h, consumed sequentially per primary sample
// (draw n of sample key K is mirt_rng_bits(K, n)); see oracle_hook_next_word below.
//
// Sub-commands (all paths relative to the directory given with --root, which must hold
// Textures/ because the reference opens "Textures/<name>" relative to the CWD):
//   dump-scene <scene>                 canonical text dump of the parsed Scene   -> stdout
//   dump-bvh   <scene>                 preorder dump of the reference BVH        -> stdout
//   hits   <scene> W H out.bin         closest hit of pixel-centre rays (time 0.5)
//   trace  <scene> W H out.bin [opts]  Raytracer::trace() of pixel-centre rays (linear RGB)
//   render <scene> W H SPP out.bin [opts]  the render loop of Raytracer.cpp:526-571 re-typed here so
//                                      that the hooked stream restarts at every primary sample
//                                      (post-gamma RGB, as Image.pixels)
//   render-real <scene> W H SPP out.bin [opts]  Raytracer::render itself (Raytracer.cpp:475-607) on one
//                                      thread; the hooked stream runs on through the whole frame
#define private public
#include "Render/Raytracer.h"
#undef private

#include <omp.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "../include/mirt_rng.h"
#include "Core/Config.h"
#include "Geometry/Bvh.h"
#include "IO/SceneUtils.h"
#include "Utils/logger.h"

extern thread_local long long g_intersection_tests;

// ---- hooked random stream ------------------------------------------------------------
static thread_local uint64_t t_key = 0;
static thread_local uint32_t t_n = 0;
static thread_local int t_phase = 0;
static thread_local uint64_t t_x = 0;

static void stream_begin(uint64_t key) {
  t_key = key;
  t_n = 0;
  t_phase = 0;
}
static double stream_draw() {  // what the harness itself draws (jitter, time)
  return (double)(mirt_rng_bits(t_key, t_n++) >> 11) * (1.0 / 9007199254740992.0);
}
extern "C" uint32_t oracle_hook_next_word() {
  if (t_phase == 0) {
    t_x = (mirt_rng_bits(t_key, t_n++) >> 11) << 11;
    t_phase = 1;
    return (uint32_t)(t_x & 0xffffffffu);
  }
  t_phase = 0;
  return (uint32_t)(t_x >> 32);
}

// ---- helpers -------------------------------------------------------------------------
static void p3(const char *k, const Vec3 &v) { std::printf("%s %.17g %.17g %.17g\n", k, v.x, v.y, v.z); }
static void p1(const char *k, double v) { std::printf("%s %.17g\n", k, v); }
static void pi(const char *k, long v) { std::printf("%s %ld\n", k, v); }
static void ps(const char *k, const std::string &v) { std::printf("%s [%s]\n", k, v.c_str()); }
static void pm(const char *k, const Mat4 &m) {
  std::printf("%s", k);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) std::printf(" %.17g", m.m[i][j]);
  std::printf("\n");
}
static void dump_material(const Material &m) {
  p3("  diffuse", m.diffuse_color);
  p3("  specular", m.specular_color);
  p3("  ambient", m.ambient_color);
  p1("  shininess", m.shininess);
  p1("  glossiness", m.glossiness);
  p1("  reflectivity", m.reflectivity);
  p1("  transparency", m.transparency);
  p1("  refractive_index", m.refractive_index);
  ps("  texture_file", m.texture_file);
  pi("  has_texture", m.has_texture);
  p3("  emission", m.emission_color);
  p1("  emission_strength", m.emission_strength);
  ps("  normal_map", m.normal_map);
  ps("  bump_map", m.bump_map);
  p1("  bump_strength", m.bump_strength);
}
static void dump_shape(const Shape &s) {
  ps("  name", s.name);
  pi("  visible", s.visible);
  pi("  has_motion", s.has_motion);
  pm("  start", s.start_transform);
  pm("  end", s.end_transform);
  pm("  o2w", s.cached_transform.object_to_world);
  pm("  w2o", s.cached_transform.world_to_object);
  dump_material(s.material);
}

static int cmd_dump_scene(const Scene &sc) {
  p3("background_color", sc.settings.background_color);
  p1("background_strength", sc.settings.background_strength);
  pi("cameras", sc.cameras.size());
  for (auto &c : sc.cameras) {
    ps("camera", c.name);
    p3("  location", c.location);
    p3("  gaze", c.gaze_direction);
    p3("  up", c.up_direction);
    p1("  focal", c.focal_length);
    p1("  sensor_w", c.sensor_width);
    p1("  sensor_h", c.sensor_height);
    pi("  res_x", c.film_resolution_x);
    pi("  res_y", c.film_resolution_y);
    pi("  dof_enabled", c.dof_enabled);
    p1("  focus_distance", c.focus_distance);
    p1("  aperture_fstop", c.aperture_fstop);
  }
  pi("lights", sc.lights.size());
  for (auto &l : sc.lights) {
    ps("light", l.name);
    p3("  location", l.location);
    p1("  intensity", l.intensity);
    p3("  color", l.color);
    ps("  light_type", l.light_type);
    ps("  area_shape", l.area_shape);
    p1("  area_size_x", l.area_size_x);
    p1("  area_size_y", l.area_size_y);
    pi("  samples", l.samples);
    p3("  normal", l.normal);
  }
  pi("spheres", sc.spheres.size());
  for (auto &s : sc.spheres) {
    dump_shape(s);
    p3("  location", s.location);
    p3("  rotation", s.rotation);
    p3("  scale", s.scale);
  }
  pi("cubes", sc.cubes.size());
  for (auto &s : sc.cubes) {
    dump_shape(s);
    p3("  location", s.translation);
    p3("  rotation", s.rotation);
    p3("  scale", s.scale);
  }
  pi("planes", sc.planes.size());
  for (auto &s : sc.planes) {
    dump_shape(s);
    pi("  points", s.points.size());
    for (auto &q : s.points) p3("   pt", q);
  }
  pi("toruses", sc.toruses.size());
  for (auto &s : sc.toruses) {
    dump_shape(s);
    p3("  location", s.location);
    p3("  rotation", s.rotation);
    p3("  scale", s.scale);
    p1("  major_radius", s.major_radius);
    p1("  minor_radius", s.minor_radius);
  }
  pi("cylinders", sc.cylinders.size());
  for (auto &s : sc.cylinders) {
    dump_shape(s);
    p3("  location", s.location);
    p3("  rotation", s.rotation);
    p3("  scale", s.scale);
    p1("  radius", s.radius);
    p1("  depth", s.depth);
  }
  pi("cones", sc.cones.size());
  for (auto &s : sc.cones) {
    dump_shape(s);
    p3("  location", s.location);
    p3("  rotation", s.rotation);
    p3("  scale", s.scale);
    p1("  radius", s.radius);
    p1("  depth", s.depth);
  }
  return 0;
}

static void dump_bvh_node(const BVHNode *n, int depth) {
  std::printf("node depth=%d leaf=%d min %.17g %.17g %.17g max %.17g %.17g %.17g idx", depth,
              (int)n->is_leaf(), n->bbox.min.x, n->bbox.min.y, n->bbox.min.z, n->bbox.max.x,
              n->bbox.max.y, n->bbox.max.z);
  if (n->is_leaf())
    for (int i : n->object_indices) std::printf(" %d", i);
  std::printf("\n");
  if (!n->is_leaf()) {
    dump_bvh_node(n->left.get(), depth + 1);
    dump_bvh_node(n->right.get(), depth + 1);
  }
}

// global primitive index (order of Bvh.cpp:244-304) of the object a HitRecord names
struct NameIndex {
  std::multimap<std::string, int> by_name;
  explicit NameIndex(const Scene &sc) {
    int i = 0;
    for (auto &s : sc.spheres) by_name.insert({s.name, i++});
    for (auto &s : sc.cubes) by_name.insert({s.name, i++});
    for (auto &s : sc.planes) by_name.insert({s.name, i++});
    for (auto &s : sc.toruses) by_name.insert({s.name, i++});
    for (auto &s : sc.cylinders) by_name.insert({s.name, i++});
    for (auto &s : sc.cones) by_name.insert({s.name, i++});
  }
  int find(const Scene &sc, const Ray &ray, const HitRecord &hit) const {
    auto r = by_name.equal_range(hit.object_name);
    if (r.first == r.second) return -2;
    auto nx = r.first;
    ++nx;
    if (nx == r.second) return r.first->second;
    // duplicate names: the object whose own intersection gives this very t
    for (auto it = r.first; it != r.second; ++it) {
      HitRecord tmp;
      if (intersect_obj(ray, sc, it->second, tmp, 1e-5, std::numeric_limits<double>::max()) &&
          tmp.t == hit.t)
        return it->second;
    }
    return r.first->second;
  }
};

template <class T> static void wr(FILE *f, const std::vector<T> &v) {
  if (!v.empty() && std::fwrite(v.data(), sizeof(T), v.size(), f) != v.size()) {
    std::perror("fwrite");
    std::exit(2);
  }
}

struct Opts {
  uint64_t seed = 0;
  int spp = 1;
};

static int usage() {
  std::fprintf(stderr,
               "usage: ref_harness --root DIR <dump-scene|dump-bvh|hits|trace|render> <scene> [W H [SPP] out] "
               "[--max-depth n] [--shadow-samples n] [--glossy n] [--motion-blur 0|1] [--textures 0|1] "
               "[--dof a d | --no-dof] [--light-intensity f] [--ambient f] [--seed n] "
               "[--tone-mapping none|reinhard|exposure] [--exposure f] [--gamma-correction 0|1]\n");
  return 1;
}

int main(int argc, char **argv) {
  std::vector<std::string> pos;
  Opts o;
  bool dof_set = false;
  double dof_a = 0, dof_d = 10;
  const char *root = nullptr;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto need = [&](int n) {
      if (i + n >= argc) std::exit(usage());
    };
    if (a == "--root") { need(1); root = argv[++i]; }
    else if (a == "--max-depth") { need(1); g_config.max_ray_depth = std::atoi(argv[++i]); }
    else if (a == "--shadow-samples") { need(1); g_config.shadow_samples = std::atoi(argv[++i]); }
    else if (a == "--glossy") { need(1); g_config.glossy_samples = std::atoi(argv[++i]); }
    else if (a == "--motion-blur") { need(1); g_config.enable_motion_blur = std::atoi(argv[++i]) != 0; }
    else if (a == "--textures") { need(1); g_config.enable_textures = std::atoi(argv[++i]) != 0; }
    else if (a == "--dof") { need(2); dof_a = std::atof(argv[++i]); dof_d = std::atof(argv[++i]); dof_set = true; }
    else if (a == "--no-dof") { dof_a = 0; dof_set = true; }
    else if (a == "--light-intensity") { need(1); g_config.light_intensity_factor = std::atof(argv[++i]); }
    else if (a == "--ambient") { need(1); g_config.ambient_factor = std::atof(argv[++i]); }
    else if (a == "--seed") { need(1); o.seed = std::strtoull(argv[++i], nullptr, 10); }
    else if (a == "--tone-mapping") { need(1); g_config.tone_mapping_mode = argv[++i]; }
    else if (a == "--exposure") { need(1); g_config.exposure = std::atof(argv[++i]); }
    else if (a == "--gamma-correction") { need(1); g_config.enable_gamma_correction = std::atoi(argv[++i]) != 0; }
    else pos.push_back(a);
  }
  if (pos.size() < 2) return usage();
  if (root && chdir(root) != 0) {
    std::perror("chdir");
    return 1;
  }
  Utils::Logger::instance().SetLevel(Utils::Level::Fatal);

  const std::string cmd = pos[0];
  Scene scene = load_scene(pos[1]);
  if (cmd == "dump-scene") return cmd_dump_scene(scene);

  if (scene.cameras.empty()) {
    std::fprintf(stderr, "no camera in %s\n", pos[1].c_str());
    return 1;
  }
  Raytracer rt(scene);
  if (cmd == "dump-bvh") {
    dump_bvh_node(rt.bvh_root.get(), 0);
    return 0;
  }

  // camera override exactly as Main.cpp:145-176
  Camera cam = scene.cameras[0];
  if (dof_set) {
    if (dof_a > 0.0) {
      cam.dof_enabled = true;
      cam.aperture_fstop = dof_a;
      cam.focus_distance = dof_d;
    } else {
      cam.dof_enabled = false;
    }
  }

  if (cmd == "hits" && pos.size() == 5) {
    // the camera of this mode is always a pinhole: pixel-centre rays carry no randomness
    cam.dof_enabled = false;
    const int W = std::atoi(pos[2].c_str()), H = std::atoi(pos[3].c_str());
    const size_t N = (size_t)W * H;
    NameIndex names(scene);
    std::vector<int32_t> ids(N, -1), front(N, 0);
    std::vector<double> t(N, 0.0), nrm(3 * N, 0.0), pt(3 * N, 0.0), uv(2 * N, 0.0);
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const size_t i = (size_t)y * W + x;
        Ray ray = cam.get_ray(x + 0.5, y + 0.5, W, H, 0.5);
        HitRecord hit;
        double tmax = std::numeric_limits<double>::max();
        if (intersect_bvh(ray, rt.bvh_root.get(), scene, hit, 1e-5, tmax)) {
          ids[i] = names.find(scene, ray, hit);
          front[i] = hit.front_face;
          t[i] = hit.t;
          for (int k = 0; k < 3; k++) {
            nrm[3 * i + k] = hit.normal[k];
            pt[3 * i + k] = hit.intersection_point[k];
          }
          uv[2 * i] = hit.u;
          uv[2 * i + 1] = hit.v;
        }
      }
    FILE *f = std::fopen(pos[4].c_str(), "wb");
    if (!f) { std::perror("fopen"); return 1; }
    wr(f, ids); wr(f, front); wr(f, t); wr(f, nrm); wr(f, pt); wr(f, uv);
    std::fclose(f);
    return 0;
  }

  if (cmd == "trace" && pos.size() == 5) {
    const int W = std::atoi(pos[2].c_str()), H = std::atoi(pos[3].c_str());
    const size_t N = (size_t)W * H;
    std::vector<double> rgb(3 * N);
    g_intersection_tests = 0;
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const size_t i = (size_t)y * W + x;
        stream_begin(mirt_rng_sample_key(o.seed, (uint32_t)i, 0));
        Ray ray = cam.get_ray(x + 0.5, y + 0.5, W, H, 0.5);
        Color c = rt.trace(ray, 0);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
      }
    FILE *f = std::fopen(pos[4].c_str(), "wb");
    if (!f) { std::perror("fopen"); return 1; }
    wr(f, rgb);
    std::fclose(f);
    std::fprintf(stderr, "prim_tests %lld\n", g_intersection_tests);
    return 0;
  }

  if (cmd == "render" && pos.size() == 6) {
    const int W = std::atoi(pos[2].c_str()), H = std::atoi(pos[3].c_str());
    const int spp = std::max(1, std::atoi(pos[4].c_str()));
    const size_t N = (size_t)W * H;
    std::vector<double> rgb(3 * N);
    g_intersection_tests = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const size_t i = (size_t)y * W + x;
        Color px(0, 0, 0);
        for (int s = 0; s < spp; s++) {                 // Raytracer.cpp:532-543
          stream_begin(mirt_rng_sample_key(o.seed, (uint32_t)i, (uint32_t)s));
          double u = x + stream_draw();
          double v = y + stream_draw();
          double time = g_config.enable_motion_blur ? stream_draw() : 0.5;
          Ray ray = cam.get_ray(u, v, W, H, time);
          px += rt.trace(ray, 0);
        }
        px /= spp;                                      // Raytracer.cpp:544
        if (g_config.enable_gamma_correction) {         // Raytracer.cpp:563-568 (tone mapping "none")
          const double gi = 1.0 / g_config.gamma;
          px.x = std::pow(px.x, gi); px.y = std::pow(px.y, gi); px.z = std::pow(px.z, gi);
        }
        rgb[3 * i] = px.x; rgb[3 * i + 1] = px.y; rgb[3 * i + 2] = px.z;
      }
    auto t1 = std::chrono::steady_clock::now();
    FILE *f = std::fopen(pos[5].c_str(), "wb");
    if (!f) { std::perror("fopen"); return 1; }
    wr(f, rgb);
    std::fclose(f);
    std::fprintf(stderr, "prim_tests %lld seconds %.6f\n", g_intersection_tests,
                 std::chrono::duration<double>(t1 - t0).count());
    return 0;
  }
  if (cmd == "render-real" && pos.size() == 6) {
    // The reference's own render loop, untouched.  One OpenMP thread, so rows run in order and the
    // two hooked generators (thread_local in the reference) consume one stream front to back.
    const int W = std::atoi(pos[2].c_str()), H = std::atoi(pos[3].c_str());
    const int spp = std::max(1, std::atoi(pos[4].c_str()));
    setenv("OMP_NUM_THREADS", "1", 1);  // so that Raytracer.cpp:479-485 keeps the thread count set here
    omp_set_num_threads(1);
    stream_begin(mirt_rng_sample_key(o.seed, 0, 0));
    g_intersection_tests = 0;
    FILE *keep = std::freopen("/dev/null", "w", stdout);  // the progress bar
    (void)keep;
    Image img = rt.render(cam, W, H, spp);
    std::vector<double> rgb(3 * (size_t)W * H);
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const size_t i = (size_t)y * W + x;
        rgb[3 * i] = img.pixels[y][x].x; rgb[3 * i + 1] = img.pixels[y][x].y; rgb[3 * i + 2] = img.pixels[y][x].z;
      }
    FILE *f = std::fopen(pos[5].c_str(), "wb");
    if (!f) { std::perror("fopen"); return 1; }
    wr(f, rgb);
    std::fclose(f);
    std::fprintf(stderr, "prim_tests %lld\n", g_intersection_tests);
    return 0;
  }
  return usage();
}
