// Command line of the drop-in renderer: the flags, defaults and exit codes of the reference's
// `raytracer` binary (Code/Core/Main.cpp:19-207), plus --seed, --device and --binary-ppm for the engine.
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <string>

#include "host_log.h"
#include "mirt_scene.h"

namespace {

void print_help(const char *prog) {
  std::cout << "Usage: " << prog << " [options]\n"
            << "Options:\n"
            << "  --scene <filename>       Input ASCII scene file\n"
            << "  --output <filename>      Specify output PPM file (default: " << g_config.output_file << ")\n"
            << "  --resolution <w> <h>     Set output image resolution (overrides scene)\n"
            << "  --samples <n>            Number of AA samples per pixel (default: " << g_config.aa_samples << ")\n"
            << "  --max-depth <n>          Maximum ray recursion depth (default: " << g_config.max_ray_depth << ")\n"
            << "  --enable-textures        Enable texture mapping (default)\n"
            << "  --disable-textures       Disable texture mapping\n"
            << "  --soft-shadows <n>       Enable soft shadows with n samples\n"
            << "  --glossy-reflection <n>  Enable glossy reflections with n samples\n"
            << "  --motion-blur <n>        Enable motion blur with n temporal samples (0 to disable)\n"
            << "  --disable-motion-blur    Disable motion blur completely\n"
            << "  --depth-of-field <a> <d> Enable DOF with aperture f-stop a and focal distance d\n"
            << "  --disable-dof            Disable depth of field\n"
            << "  --light-intensity <f>    Global light intensity multiplier (default: "
            << g_config.light_intensity_factor << ")\n"
            << "  --ambient-light <f>      Ambient light factor (default: " << g_config.ambient_factor << ")\n"
            << "  --threads <n>            Number of rendering threads (accepted, unused)\n"
            << "  --log-level <level>      Set log level (debug, info, warn, error)\n"
            << "  --seed <n>               Seed of the sampling sequence (default: 0)\n"
            << "  --device <n>             HIP device to render on (default: 0)\n"
            << "  --devices <n>            Split every frame over n GPUs, device .. device + n - 1 (default: 1)\n"
            << "  --gpu-bvh                Build the BVH on the GPU (large scenes without planes)\n"
            << "  --binary-ppm             Write the image as binary PPM (P6) instead of ASCII (P3)\n"
            << "  --help                   Display this help message\n";
}

void parse_arguments(int argc, char *argv[]) {
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    const int left = argc - 1 - i;  // arguments after this one
    if (a == "--help") { print_help(argv[0]); std::exit(0); }
    else if ((a == "--scene" || a == "-s") && left >= 1) g_config.scene_file = argv[++i];
    else if ((a == "--output" || a == "-o") && left >= 1) g_config.output_file = argv[++i];
    else if (a == "--resolution" && left >= 2) {
      g_config.override_width = std::atoi(argv[++i]);
      g_config.override_height = std::atoi(argv[++i]);
    }
    else if ((a == "-w" || a == "-W") && left >= 1) g_config.override_width = std::atoi(argv[++i]);
    else if (a == "-H" && left >= 1) g_config.override_height = std::atoi(argv[++i]);
    else if (a == "--samples" && left >= 1) {
      g_config.aa_samples = std::atoi(argv[++i]);
      if (g_config.aa_samples < 1) g_config.aa_samples = 1;
    }
    else if (a == "--max-depth" && left >= 1) g_config.max_ray_depth = std::atoi(argv[++i]);
    else if (a == "--enable-textures") g_config.enable_textures = true;
    else if (a == "--disable-textures") g_config.enable_textures = false;
    else if (a == "--soft-shadows" && left >= 1) g_config.shadow_samples = std::atoi(argv[++i]);
    else if (a == "--glossy-reflection" && left >= 1) g_config.glossy_samples = std::atoi(argv[++i]);
    else if (a == "--motion-blur" && left >= 1) {
      g_config.motion_blur_samples = std::atoi(argv[++i]);
      g_config.enable_motion_blur = g_config.motion_blur_samples > 0;
    }
    else if (a == "--disable-motion-blur") g_config.enable_motion_blur = false;
    else if (a == "--depth-of-field" && left >= 2) {
      g_config.lens_aperture = std::atof(argv[++i]);
      g_config.lens_focal_distance = std::atof(argv[++i]);
      g_config.dof_flag_set = true;
    }
    else if (a == "--disable-dof") { g_config.lens_aperture = 0.0; g_config.dof_flag_set = true; }
    else if (a == "--light-intensity" && left >= 1) g_config.light_intensity_factor = std::atof(argv[++i]);
    else if (a == "--ambient-light" && left >= 1) g_config.ambient_factor = std::atof(argv[++i]);
    else if (a == "--threads" && left >= 1) g_config.num_threads = std::atoi(argv[++i]);
    else if (a == "--log-level" && left >= 1) g_config.log_level = argv[++i];
    else if (a == "--seed" && left >= 1) g_config.seed = std::strtoull(argv[++i], nullptr, 10);
    else if (a == "--device" && left >= 1) g_config.device = std::atoi(argv[++i]);
    else if (a == "--devices" && left >= 1) g_config.devices = std::max(1, std::atoi(argv[++i]));
    else if (a == "--gpu-bvh") g_config.gpu_bvh = true;
    else if (a == "--binary-ppm") g_config.output_format = "p6";
    else {
      host_log(LogLevel::Error, "Unknown argument arg=%s", a.c_str());
      print_help(argv[0]);
      std::exit(1);
    }
  }
}

}  // namespace

int main(int argc, char *argv[]) {
  if (argc <= 1) {
    print_help(argv[0]);
    return 0;
  }
  parse_arguments(argc, argv);
  // kernel arguments in device memory (the HIP runtime reads this when it starts, which is at the first render)
  setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
  if (g_config.scene_file.empty()) {
    host_log(LogLevel::Error, "No scene file specified");
    print_help(argv[0]);
    return 1;
  }
  host_log_set_level(g_config.log_level);

  Scene scene = load_scene(g_config.scene_file);
  host_log(LogLevel::Info, "Loaded scene path=%s objects=%zu lights=%zu", g_config.scene_file.c_str(),
           scene.object_count(), scene.lights.size());
  if (scene.cameras.empty()) {
    host_log(LogLevel::Error, "Scene must contain at least one camera");
    return 1;
  }

  Camera cam = scene.cameras[0];  // only the first camera renders
  int width = cam.film_resolution_x, height = cam.film_resolution_y;
  if (g_config.override_width > 0 && g_config.override_height > 0) {
    width = g_config.override_width;
    height = g_config.override_height;
    host_log(LogLevel::Info, "Overriding resolution width=%d height=%d", width, height);
  }
  if (g_config.dof_flag_set) {
    if (g_config.lens_aperture > 0.0) {
      cam.dof_enabled = true;
      cam.aperture_fstop = g_config.lens_aperture;
      cam.focus_distance = g_config.lens_focal_distance;
    } else {
      cam.dof_enabled = false;
    }
  }
  host_log(LogLevel::Info,
           "Render Configuration width=%d height=%d aa_samples=%d shadow_samples=%d glossy_samples=%d "
           "max_ray_depth=%d textures=%d motion_blur=%d",
           width, height, g_config.aa_samples, g_config.shadow_samples, g_config.glossy_samples,
           g_config.max_ray_depth, static_cast<int>(g_config.enable_textures),
           static_cast<int>(g_config.enable_motion_blur));

  try {
    Image out =
        render_scene_bvh_antialiased(scene, cam, width, height, g_config.aa_samples, g_config.shadow_samples);
    if (g_config.output_format == "p6") out.magic_number = "P6";  // a 4K frame is 25 MB instead of ~90 MB of text
    out.write(g_config.output_file);
  } catch (const std::exception &e) {
    host_log(LogLevel::Fatal, "%s", e.what());
    return 2;
  }
  host_log(LogLevel::Info, "Image saved path=%s", g_config.output_file.c_str());
  return 0;
}
