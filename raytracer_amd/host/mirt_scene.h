// The drop-in host surface: the scene model, configuration, image type and entry points that host
// code written against the reference renderer uses, with the same names, fields and defaults
// (Code/Core/{Ray,Material,Light,Camera,Scene,Config}.h, Code/Geometry/{Shape,Sphere,Cube,Plane,
// Torus,Cylinder,Cone}.h, Code/IO/{Image,SceneUtils}.h, Code/Render/Raytracer.h).  Behind
// Raytracer::render the work is done by the HIP engine through the C ABI of include/mirt.h; nothing
// here traces a ray.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "mirt_math.h"

struct Ray {
  Point origin;
  Direction direction;
  double time = 0.5;
  Ray() = default;
  Ray(const Point &o, const Direction &d, double t = 0.5) : origin(o), direction(d), time(t) {}
};

struct Material {
  Color diffuse_color{0.8, 0.8, 0.8};
  Color specular_color{1.0, 1.0, 1.0};
  Color ambient_color{0.1, 0.1, 0.1};
  double shininess{32.0};
  double glossiness{0.0};
  double reflectivity{0.0};
  double transparency{0.0};
  double refractive_index{1.0};
  std::string texture_file;
  bool has_texture{false};
  Color emission_color{0.0, 0.0, 0.0};
  double emission_strength{0.0};
  double subsurface{0.0};
  double sheen{0.0};
  double clearcoat{0.0};
  double clearcoat_roughness{0.0};
  std::string normal_map;
  std::string bump_map;
  double bump_strength{1.0};
};

struct Light {
  std::string name;
  Point location;
  double intensity = 0.0;
  Color color;
  std::string light_type{"POINT"};
  double spot_size{0.785398};
  double spot_blend{0.15};
  std::string area_shape{"SQUARE"};
  double area_size_x{1.0};
  double area_size_y{1.0};
  int samples{16};
  Direction normal{0.0, 0.0, -1.0};
  Direction direction{0.0, 0.0, -1.0};
  double angle{0.0};
  bool cast_shadows{true};
  double shadow_soft_size{0.0};
};

struct Camera {
  std::string name;
  Point location;
  Direction gaze_direction;
  Direction up_direction;
  double focal_length = 0.0;
  double sensor_width = 0.0;
  double sensor_height = 0.0;
  int film_resolution_x = 0;
  int film_resolution_y = 0;
  bool dof_enabled{false};
  double focus_distance{10.0};
  double aperture_fstop{2.8};
  int aperture_blades{0};
  std::string camera_type{"PERSP"};
  double clip_start{0.1};
  double clip_end{100.0};
};

struct Shape {
  std::string name;
  std::string type;
  Material material;
  bool visible{true};
  bool has_motion{false};
  Mat4 start_transform;
  Mat4 end_transform;
  Transform cached_transform;
};

struct Sphere : Shape {
  Point location;
  Vec3 scale{1.0, 1.0, 1.0};
  Point rotation{0.0, 0.0, 0.0};
};
struct Cube : Shape {
  Point translation;
  Point rotation;
  Vec3 scale{1.0, 1.0, 1.0};
};
struct Plane : Shape {
  std::vector<Point> points;
};
struct Torus : Shape {
  Point location;
  Point rotation;
  Vec3 scale{1.0, 1.0, 1.0};
  double major_radius{};
  double minor_radius{};
};
struct Cylinder : Shape {
  Point location;
  Point rotation;
  Vec3 scale{1.0, 1.0, 1.0};
  double radius{};
  double depth{};
};
struct Cone : Shape {
  Point location;
  Point rotation;
  Vec3 scale{1.0, 1.0, 1.0};
  double radius{};
  double depth{};
};

struct SceneSettings {
  Color background_color{0.05, 0.05, 0.05};
  double background_strength{1.0};
  Color ambient_light{0.1, 0.1, 0.1};
  int frame_current{1};
  int frame_start{1};
  int frame_end{250};
  int fps{24};
  int max_bounces{12};
  int diffuse_bounces{4};
  int glossy_bounces{4};
  int transmission_bounces{12};
};

struct Scene {
  SceneSettings settings;
  std::vector<Camera> cameras;
  std::vector<Light> lights;
  std::vector<Sphere> spheres;
  std::vector<Cube> cubes;
  std::vector<Plane> planes;
  std::vector<Torus> toruses;
  std::vector<Cylinder> cylinders;
  std::vector<Cone> cones;
  std::size_t object_count() const {
    return spheres.size() + cubes.size() + planes.size() + toruses.size() + cylinders.size() + cones.size();
  }
};

// Every tunable of the reference's RenderConfig (Config.h:11-168), same names and defaults.
struct RenderConfig {
  int aa_samples = 4;
  int shadow_samples = 16;
  int max_ray_depth = 12;
  std::string tone_mapping_mode = "none";
  double exposure = 1.0;
  double reinhard_white_point = 10.0;
  double gamma = 2.2;
  bool enable_gamma_correction = true;
  double light_intensity_factor = 0.2;
  double ambient_factor = 1.0;
  double ray_offset_epsilon = 0.001;
  bool use_adaptive_epsilon = true;
  double adaptive_epsilon_scale = 0.0001;
  bool enable_textures = true;
  std::string texture_filter = "bilinear";
  int num_threads = 0;
  bool print_stats = true;
  bool debug_mode = false;
  std::string log_level = "info";
  std::string output_format = "both";
  bool auto_convert_to_png = true;
  int dof_samples = 32;
  int motion_blur_samples = 16;
  bool enable_motion_blur = true;
  int glossy_samples = 0;
  double lens_aperture = 0.0;
  double lens_focal_distance = 10.0;
  bool dof_flag_set = false;
  std::string scene_file;
  std::string output_file = "output.ppm";
  int override_width = 0;
  int override_height = 0;
  double pure_glass_threshold = 0.99;
  double glass_absorption_strength = 0.1;
  double pure_emission_threshold = 4.0;
  // engine additions (no counterpart in the reference): seed of the addressed random function,
  // HIP device to render on, number of GPUs to split a frame over (devices device .. device + devices - 1)
  unsigned long long seed = 0;
  int device = 0;
  int devices = 1;
  // build the BVH on the GPU instead of with build_bvh (large scenes without planes; mirt_scene_create_with)
  bool gpu_bvh = false;
};
extern RenderConfig g_config;

// PPM image of doubles in [0,1] (Image.h:10-51): P3/P6 reader, P3/P6 writer.
struct Image {
  int height = 0;
  int width = 0;
  int max_value = 255;
  std::string magic_number;
  std::vector<std::vector<Pixel>> pixels;
  Image(int height, int width, int max_value, std::string magic_number);
  explicit Image(const std::string &filepath);
  void write(const std::string &filepath) const;
};

Scene load_scene(const std::string &filepath);

// Flattened, engine-ready form of a Scene (see flatten.h); opaque here.
struct FlatScene;
struct mirt_scene;
struct mirt_stats;

class Raytracer {
 public:
  explicit Raytracer(const Scene &scene);
  ~Raytracer();
  Raytracer(const Raytracer &) = delete;
  Raytracer &operator=(const Raytracer &) = delete;
  Image render(const Camera &camera, int width, int height, int samples_per_pixel);
  const mirt_stats &last_stats() const;

 private:
  const Scene &scene;
  std::unique_ptr<FlatScene> flat;
  mirt_scene *device_scene = nullptr;  // one GPU ...
  struct mirt_group *device_group = nullptr;  // ... or several (g_config.devices > 1)
  std::unique_ptr<mirt_stats> stats;
};

Image render_scene_bvh_antialiased(const Scene &scene, const Camera &camera, int width, int height,
                                   int samples_per_pixel, int shadow_samples);
