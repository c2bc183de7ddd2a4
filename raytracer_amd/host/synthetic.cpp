// Synthetic scene generator and ASCII scene writer (see synthetic.h).
#include "synthetic.h"

#include <cmath>
#include <cstdio>

#include "mirt_rng.h"

namespace {

// Random numbers of primitive i: the addressed function of mirt_rng.h keyed by (seed, i).
struct Draws {
  std::uint64_t key;
  std::uint32_t dim = 0;
  Draws(std::uint64_t seed, std::uint32_t i) : key(mirt_rng_sample_key(seed, i, 0)) {}
  double next() { return mirt_rng_draw(key, dim++); }
  double range(double lo, double hi) { return lo + (hi - lo) * next(); }
};

void fill_common(Shape &s, Draws &r, const char *type, int index, Point &location, Point &rotation) {
  s.name = std::string(type) + "_" + std::to_string(index);
  s.type = type;
  location = Point(r.range(-100.0, 100.0), r.range(-100.0, 100.0), r.range(-100.0, 100.0));
  rotation = Point(r.range(0.0, 2.0 * M_PI), r.range(0.0, 2.0 * M_PI), r.range(0.0, 2.0 * M_PI));
  Material &m = s.material;
  const double kind = r.next();
  const Color tint(r.range(0.2, 1.0), r.range(0.2, 1.0), r.range(0.2, 1.0));
  m.diffuse_color = tint;
  m.ambient_color = tint * 0.1;
  m.specular_color = Color(0.5, 0.5, 0.5);
  m.shininess = 64.0;
  m.refractive_index = 1.0;
  if (kind < 0.20) {  // mirror
    m.reflectivity = 0.9;
    m.glossiness = 1.0;
  } else if (kind < 0.25) {  // glass
    m.transparency = 1.0;
    m.refractive_index = 1.45;
    m.glossiness = 1.0;
  }
}

double log_uniform_size(Draws &r) { return 0.05 * std::pow(10.0, r.next()); }

void p3(std::FILE *f, const char *key, const Vec3 &v) { std::fprintf(f, "%s %.17g %.17g %.17g\n", key, v.x, v.y, v.z); }
void p1(std::FILE *f, const char *key, double v) { std::fprintf(f, "%s %.17g\n", key, v); }

void write_material(std::FILE *f, const Shape &s) {
  const Material &m = s.material;
  std::fprintf(f, "visible %d\n", s.visible ? 1 : 0);
  p3(f, "material_diffuse", m.diffuse_color);
  p3(f, "material_specular", m.specular_color);
  p3(f, "material_ambient", m.ambient_color);
  p1(f, "material_shininess", m.shininess);
  p1(f, "material_glossiness", m.glossiness);
  p1(f, "material_reflectivity", m.reflectivity);
  p1(f, "material_transparency", m.transparency);
  p1(f, "material_refractive_index", m.refractive_index);
}

template <class S> void write_placement(std::FILE *f, const S &s) {
  std::fprintf(f, "name %s\n", s.name.c_str());
  p3(f, "location", s.location);
  p3(f, "rotation", s.rotation);
  p3(f, "scale", s.scale);
}

}  // namespace

Scene make_synthetic_scene(int n_prims, std::uint64_t seed) {
  Scene scene;
  if (n_prims < 0) n_prims = 0;
  scene.settings.background_color = Color(0.05, 0.06, 0.08);
  scene.settings.background_strength = 1.0;

  Camera cam;
  cam.name = "Main_Camera";
  cam.location = Point(0.0, -330.0, 60.0);
  cam.gaze_direction = (Point(0, 0, 0) - cam.location).norm();
  const Direction right = cam.gaze_direction.cross(Direction(0, 0, 1)).norm();
  cam.up_direction = right.cross(cam.gaze_direction).norm();
  cam.focal_length = 30.0;
  cam.sensor_width = 36.0;
  cam.sensor_height = 20.25;
  cam.film_resolution_x = 3840;
  cam.film_resolution_y = 2160;
  cam.clip_end = 1000.0;
  scene.cameras.push_back(cam);

  const Point light_at[3] = {Point(-250.0, -300.0, 250.0), Point(300.0, -200.0, 150.0), Point(0.0, 100.0, 350.0)};
  const Color light_colour[3] = {Color(1.0, 0.95, 0.9), Color(0.9, 0.95, 1.0), Color(1.0, 1.0, 1.0)};
  for (int i = 0; i < 3; ++i) {
    Light l;
    l.name = "Light_" + std::to_string(i);
    l.location = light_at[i];
    l.intensity = 400000.0;  // 0.2 * I / d^2 is of order one across the cloud
    l.color = light_colour[i];
    l.light_type = "POINT";
    scene.lights.push_back(l);
  }

  const int quarter = n_prims / 4;
  const int n_sphere = n_prims - 3 * quarter;
  scene.spheres.reserve(n_sphere);
  scene.cylinders.reserve(quarter);
  scene.cones.reserve(quarter);
  scene.toruses.reserve(quarter);
  std::uint32_t id = 0;
  for (int i = 0; i < n_sphere; ++i, ++id) {
    Draws r(seed, id);
    Sphere s;
    fill_common(s, r, "Sphere", i, s.location, s.rotation);
    const double size = log_uniform_size(r);
    s.scale = Vec3(size, size, size);
    s.cached_transform = Transform::from_trs_nonuniform(s.location, s.rotation, s.scale);
    scene.spheres.push_back(std::move(s));
  }
  for (int i = 0; i < quarter; ++i, ++id) {
    Draws r(seed, id);
    Torus s;
    fill_common(s, r, "Torus", i, s.location, s.rotation);
    const double size = log_uniform_size(r);
    s.major_radius = size;
    s.minor_radius = size * 0.3;
    s.cached_transform = Transform::from_trs_nonuniform(s.location, s.rotation, s.scale);
    scene.toruses.push_back(std::move(s));
  }
  for (int i = 0; i < quarter; ++i, ++id) {
    Draws r(seed, id);
    Cylinder s;
    fill_common(s, r, "Cylinder", i, s.location, s.rotation);
    const double size = log_uniform_size(r);
    s.radius = size;
    s.depth = 2.0 * size;
    s.cached_transform = Transform::from_trs_nonuniform(s.location, s.rotation, s.scale);
    scene.cylinders.push_back(std::move(s));
  }
  for (int i = 0; i < quarter; ++i, ++id) {
    Draws r(seed, id);
    Cone s;
    fill_common(s, r, "Cone", i, s.location, s.rotation);
    const double size = log_uniform_size(r);
    s.radius = size;
    s.depth = 2.0 * size;
    s.cached_transform = Transform::from_trs_nonuniform(s.location, s.rotation, s.scale);
    scene.cones.push_back(std::move(s));
  }
  return scene;
}

bool write_scene_ascii(const Scene &scene, const std::string &path) {
  std::FILE *f = std::fopen(path.c_str(), "w");
  if (!f) return false;
  const SceneSettings &st = scene.settings;
  std::fprintf(f, "SCENE_SETTINGS\n");
  p3(f, "background_color", st.background_color);
  p1(f, "background_strength", st.background_strength);
  p3(f, "ambient_light", st.ambient_light);
  std::fprintf(f, "frame_current %d\nframe_start %d\nframe_end %d\nfps %d\n", st.frame_current, st.frame_start,
               st.frame_end, st.fps);
  std::fprintf(f, "max_bounces %d\ndiffuse_bounces %d\nglossy_bounces %d\ntransmission_bounces %d\n\n", st.max_bounces,
               st.diffuse_bounces, st.glossy_bounces, st.transmission_bounces);
  std::fprintf(f, "CAMERAS %zu\n", scene.cameras.size());
  for (const Camera &c : scene.cameras) {
    std::fprintf(f, "name %s\n", c.name.c_str());
    p3(f, "location", c.location);
    p3(f, "gaze", c.gaze_direction);
    p3(f, "up", c.up_direction);
    p1(f, "focal", c.focal_length);
    std::fprintf(f, "sensor %.17g %.17g\n", c.sensor_width, c.sensor_height);
    std::fprintf(f, "resolution %d %d\n", c.film_resolution_x, c.film_resolution_y);
    std::fprintf(f, "dof_enabled %d\n", c.dof_enabled ? 1 : 0);
    p1(f, "focus_distance", c.focus_distance);
    p1(f, "aperture_fstop", c.aperture_fstop);
    std::fprintf(f, "aperture_blades %d\ncamera_type %s\n", c.aperture_blades, c.camera_type.c_str());
    p1(f, "clip_start", c.clip_start);
    p1(f, "clip_end", c.clip_end);
  }
  std::fprintf(f, "LIGHTS %zu\n", scene.lights.size());
  for (const Light &l : scene.lights) {
    std::fprintf(f, "name %s\n", l.name.c_str());
    p3(f, "location", l.location);
    p1(f, "intensity", l.intensity);
    p3(f, "color", l.color);
    std::fprintf(f, "light_type %s\ncast_shadows %d\n", l.light_type.c_str(), l.cast_shadows ? 1 : 0);
    p1(f, "shadow_soft_size", l.shadow_soft_size);
  }
  std::fprintf(f, "SPHERES %zu\n", scene.spheres.size());
  for (const Sphere &s : scene.spheres) {
    write_placement(f, s);
    write_material(f, s);
  }
  std::fprintf(f, "CUBES 0\nPLANES 0\n");
  std::fprintf(f, "TORUSES %zu\n", scene.toruses.size());
  for (const Torus &s : scene.toruses) {
    write_placement(f, s);
    p1(f, "major_radius", s.major_radius);
    p1(f, "minor_radius", s.minor_radius);
    write_material(f, s);
  }
  std::fprintf(f, "CYLINDERS %zu\n", scene.cylinders.size());
  for (const Cylinder &s : scene.cylinders) {
    write_placement(f, s);
    p1(f, "radius", s.radius);
    p1(f, "depth", s.depth);
    write_material(f, s);
  }
  std::fprintf(f, "CONES %zu\n", scene.cones.size());
  for (const Cone &s : scene.cones) {
    write_placement(f, s);
    p1(f, "radius", s.radius);
    p1(f, "depth", s.depth);
    write_material(f, s);
  }
  return std::fclose(f) == 0;
}
