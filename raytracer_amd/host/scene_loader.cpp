// ASCII scene reader of the drop-in surface: `Scene load_scene(path)` with the grammar and quirks
// of the reference loader (Code/IO/SceneLoader.cpp:24-110 material block, :112-904 sections,
// :868-889 transform precompute; grammar written by Blend/Export.py:587-744).
//
// The format is line oriented.  A section header (`SPHERES 2`) is followed by that many objects;
// an object is a run of POSITIONAL mandatory lines (the reference never checks their keywords, only
// their order), then an order-free block of optional lines (`visible`, `motion_blur`, `matrix_t0`,
// `matrix_t1` + 4 rows each), then `material_*` lines; each block ends at the first line whose
// keyword it does not know, which is handed back to the enclosing reader.
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "mirt_scene.h"

namespace {

// The file as a list of lines with a cursor; "unread" replaces the reference's tellg/seekg.
class Lines {
 public:
  explicit Lines(std::istream &in) {
    std::string l;
    while (std::getline(in, l)) lines_.push_back(std::move(l));
  }
  bool next(std::string &out) {
    if (at_ >= lines_.size()) return false;
    out = lines_[at_++];
    return true;
  }
  void unread() { --at_; }

 private:
  std::vector<std::string> lines_;
  std::size_t at_ = 0;
};

// One line split into its keyword and the stream positioned after it.
struct Fields {
  std::string key;
  std::istringstream rest;
  explicit Fields(const std::string &line) : rest(line) { rest >> key; }
  Fields &operator>>(double &v) { rest >> v; return *this; }
  Fields &operator>>(int &v) { rest >> v; return *this; }
  Fields &operator>>(Vec3 &v) { rest >> v.x >> v.y >> v.z; return *this; }
  bool flag() { int v = 0; rest >> v; return v != 0; }
  // remainder of the line without leading blanks (names may contain spaces)
  std::string text() {
    std::string s;
    std::getline(rest, s);
    const auto b = s.find_first_not_of(" \t");
    return b == std::string::npos ? std::string() : s.substr(b);
  }
  // remainder of the line minus exactly one separator character (enumerations)
  std::string word() {
    std::string s;
    std::getline(rest, s);
    return s.empty() ? s : s.substr(1);
  }
};

Fields take(Lines &in) {
  std::string l;
  in.next(l);
  return Fields(l);
}

void read_matrix(Lines &in, Mat4 &m, const char *what) {
  for (int r = 0; r < 4; ++r) {
    std::string l;
    if (!in.next(l)) throw std::runtime_error(std::string("Failed to read ") + what + " row");
    std::istringstream row(l);
    row >> m.m[r][0] >> m.m[r][1] >> m.m[r][2] >> m.m[r][3];
  }
}

Material read_material(Lines &in) {
  Material m;
  std::string l;
  while (in.next(l)) {
    Fields f(l);
    const std::string &k = f.key;
    if (k.empty()) continue;
    if (k == "material_diffuse") f >> m.diffuse_color;
    else if (k == "material_specular") f >> m.specular_color;
    else if (k == "material_ambient") f >> m.ambient_color;
    else if (k == "material_shininess") f >> m.shininess;
    else if (k == "material_glossiness") f >> m.glossiness;
    else if (k == "material_reflectivity") f >> m.reflectivity;
    else if (k == "material_transparency") f >> m.transparency;
    else if (k == "material_refractive_index") f >> m.refractive_index;
    else if (k == "material_texture") { m.texture_file = f.text(); m.has_texture = true; }
    else if (k == "material_emission") f >> m.emission_color;
    else if (k == "material_emission_strength") f >> m.emission_strength;
    else if (k == "material_subsurface") f >> m.subsurface;
    else if (k == "material_sheen") f >> m.sheen;
    else if (k == "material_clearcoat") f >> m.clearcoat;
    else if (k == "material_clearcoat_roughness") f >> m.clearcoat_roughness;
    else if (k == "material_normal_map") m.normal_map = f.text();
    else if (k == "material_bump_map") m.bump_map = f.text();
    else if (k == "material_bump_strength") f >> m.bump_strength;
    else { in.unread(); break; }
  }
  return m;
}

// visible / motion_blur / matrix_t0 / matrix_t1, in any order, then the material block
void read_shape_tail(Lines &in, Shape &s) {
  std::string l;
  while (in.next(l)) {
    Fields f(l);
    if (f.key.empty()) continue;
    if (f.key == "visible") s.visible = f.flag();
    else if (f.key == "motion_blur") s.has_motion = f.flag();
    else if (f.key == "matrix_t0") read_matrix(in, s.start_transform, "matrix_t0");
    else if (f.key == "matrix_t1") read_matrix(in, s.end_transform, "matrix_t1");
    else { in.unread(); break; }
  }
  s.material = read_material(in);
}

// name / location / rotation / scale, by position
void read_placement(Lines &in, Shape &s, Vec3 &loc, Vec3 &rot, Vec3 &scale) {
  s.name = take(in).text();
  take(in) >> loc;
  take(in) >> rot;
  take(in) >> scale;
}

void read_camera(Lines &in, Camera &c) {
  c.name = take(in).text();
  take(in) >> c.location;
  take(in) >> c.gaze_direction;
  take(in) >> c.up_direction;
  take(in) >> c.focal_length;
  take(in) >> c.sensor_width >> c.sensor_height;
  take(in) >> c.film_resolution_x >> c.film_resolution_y;
  std::string l;
  while (in.next(l)) {
    Fields f(l);
    const std::string &k = f.key;
    if (k.empty()) continue;
    if (k == "dof_enabled") c.dof_enabled = f.flag();
    else if (k == "focus_distance") f >> c.focus_distance;
    else if (k == "aperture_fstop") f >> c.aperture_fstop;
    else if (k == "aperture_blades") f >> c.aperture_blades;
    else if (k == "camera_type") c.camera_type = f.word();
    else if (k == "clip_start") f >> c.clip_start;
    else if (k == "clip_end") f >> c.clip_end;
    else { in.unread(); break; }
  }
}

void read_light(Lines &in, Light &g) {
  g.name = take(in).text();
  take(in) >> g.location;
  take(in) >> g.intensity;
  take(in) >> g.color;
  std::string l;
  while (in.next(l)) {
    Fields f(l);
    const std::string &k = f.key;
    if (k.empty()) continue;
    if (k == "light_type") g.light_type = f.word();
    else if (k == "spot_size") f >> g.spot_size;
    else if (k == "spot_blend") f >> g.spot_blend;
    else if (k == "area_shape") g.area_shape = f.word();
    else if (k == "area_size") f >> g.area_size_x >> g.area_size_y;
    else if (k == "direction") f >> g.direction;
    else if (k == "angle") f >> g.angle;
    else if (k == "cast_shadows") g.cast_shadows = f.flag();
    else if (k == "shadow_soft_size") f >> g.shadow_soft_size;
    else if (k == "samples") f >> g.samples;
    else if (k == "normal") f >> g.normal;
    else { in.unread(); break; }
  }
}

bool is_section(const std::string &k) {
  return k == "CAMERAS" || k == "LIGHTS" || k == "SPHERES" || k == "CUBES" || k == "PLANES" || k == "TORUSES" ||
         k == "CYLINDERS" || k == "CONES";
}

void read_settings(Lines &in, SceneSettings &s) {
  std::string l;
  while (in.next(l)) {
    Fields f(l);
    const std::string &k = f.key;
    if (k == "background_color") f >> s.background_color;
    else if (k == "background_strength") f >> s.background_strength;
    else if (k == "ambient_light") f >> s.ambient_light;
    else if (k == "frame_current") f >> s.frame_current;
    else if (k == "frame_start") f >> s.frame_start;
    else if (k == "frame_end") f >> s.frame_end;
    else if (k == "fps") f >> s.fps;
    else if (k == "max_bounces") f >> s.max_bounces;
    else if (k == "diffuse_bounces") f >> s.diffuse_bounces;
    else if (k == "glossy_bounces") f >> s.glossy_bounces;
    else if (k == "transmission_bounces") f >> s.transmission_bounces;
    else if (is_section(k)) { in.unread(); break; }
  }
}

}  // namespace

Scene load_scene(const std::string &filepath) {
  Scene scene;
  std::ifstream file(filepath);
  if (!file.is_open()) {
    std::cerr << "[ERROR] Failed to open scene file file=" << filepath << "\n";
    return scene;  // an empty scene, as the reference returns (SceneLoader.cpp:116-122)
  }
  Lines in(file);
  std::string l;
  while (in.next(l)) {
    Fields head(l);
    const std::string &k = head.key;
    int count = 0;
    if (k == "SCENE_SETTINGS") {
      read_settings(in, scene.settings);
    } else if (k == "CAMERAS") {
      head >> count;
      for (int i = 0; i < count; ++i) { Camera c; read_camera(in, c); scene.cameras.push_back(std::move(c)); }
    } else if (k == "LIGHTS") {
      head >> count;
      for (int i = 0; i < count; ++i) { Light g; read_light(in, g); scene.lights.push_back(std::move(g)); }
    } else if (k == "SPHERES") {
      head >> count;
      scene.spheres.reserve(count > 0 ? count : 0);
      for (int i = 0; i < count; ++i) {
        Sphere s; s.type = "sphere";
        read_placement(in, s, s.location, s.rotation, s.scale);
        read_shape_tail(in, s);
        scene.spheres.push_back(std::move(s));
      }
    } else if (k == "CUBES") {
      head >> count;
      scene.cubes.reserve(count > 0 ? count : 0);
      for (int i = 0; i < count; ++i) {
        Cube s; s.type = "cube";
        read_placement(in, s, s.translation, s.rotation, s.scale);
        read_shape_tail(in, s);
        scene.cubes.push_back(std::move(s));
      }
    } else if (k == "PLANES") {
      head >> count;
      scene.planes.reserve(count > 0 ? count : 0);
      for (int i = 0; i < count; ++i) {
        Plane s; s.type = "plane";
        s.name = take(in).text();
        int n = 0;
        take(in) >> n;
        for (int j = 0; j < n; ++j) {
          std::string pl;
          in.next(pl);
          std::istringstream ps(pl);
          Point p;
          ps >> p.x >> p.y >> p.z;
          s.points.push_back(p);
        }
        read_shape_tail(in, s);
        scene.planes.push_back(std::move(s));
      }
    } else if (k == "TORUSES") {
      head >> count;
      scene.toruses.reserve(count > 0 ? count : 0);
      for (int i = 0; i < count; ++i) {
        Torus s; s.type = "torus";
        read_placement(in, s, s.location, s.rotation, s.scale);
        take(in) >> s.major_radius;
        take(in) >> s.minor_radius;
        read_shape_tail(in, s);
        scene.toruses.push_back(std::move(s));
      }
    } else if (k == "CYLINDERS") {
      head >> count;
      scene.cylinders.reserve(count > 0 ? count : 0);
      for (int i = 0; i < count; ++i) {
        Cylinder s; s.type = "cylinder";
        read_placement(in, s, s.location, s.rotation, s.scale);
        take(in) >> s.radius;
        take(in) >> s.depth;
        read_shape_tail(in, s);
        scene.cylinders.push_back(std::move(s));
      }
    } else if (k == "CONES") {
      head >> count;
      scene.cones.reserve(count > 0 ? count : 0);
      for (int i = 0; i < count; ++i) {
        Cone s; s.type = "cone";
        read_placement(in, s, s.location, s.rotation, s.scale);
        take(in) >> s.radius;
        take(in) >> s.depth;
        read_shape_tail(in, s);
        scene.cones.push_back(std::move(s));
      }
    }
  }

  // object-to-world = T * Rz * Ry * Rx * S and its inverse, once per shape; planes keep the
  // identity (their points are already in world space)
  for (auto &s : scene.spheres) s.cached_transform = Transform::from_trs_nonuniform(s.location, s.rotation, s.scale);
  for (auto &s : scene.cubes) s.cached_transform = Transform::from_trs_nonuniform(s.translation, s.rotation, s.scale);
  for (auto &s : scene.toruses) s.cached_transform = Transform::from_trs_nonuniform(s.location, s.rotation, s.scale);
  for (auto &s : scene.cylinders) s.cached_transform = Transform::from_trs_nonuniform(s.location, s.rotation, s.scale);
  for (auto &s : scene.cones) s.cached_transform = Transform::from_trs_nonuniform(s.location, s.rotation, s.scale);
  return scene;
}
