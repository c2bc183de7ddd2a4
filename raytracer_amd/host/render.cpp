// `Raytracer` and `render_scene_bvh_antialiased` of the drop-in surface
// (Code/Render/Raytracer.h:24-67).  The constructor does what the reference constructor does --
// number the objects, build the BVH, preload textures (Raytracer.cpp:26-71) -- then hands the flat
// scene to the HIP engine; render() snapshots g_config, asks the engine for the frame and copies
// it into an Image exactly as the reference fills Image.pixels (Raytracer.cpp:570).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "flatten.h"
#include "host_log.h"
#include "mirt.h"
#include "mirt_scene.h"

RenderConfig g_config;

Raytracer::Raytracer(const Scene &s) : scene(s), stats(new mirt_stats) {
  std::memset(stats.get(), 0, sizeof(mirt_stats));
  host_log(LogLevel::Info, "Initializing Raytracer and building BVH...");
  const auto t0 = std::chrono::steady_clock::now();
  flat.reset(new FlatScene(flatten_scene(scene, "", !g_config.gpu_bvh)));  // (--gpu-bvh: the engine builds the tree)
  const auto ms =
      std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  host_log(LogLevel::Info, "BVH construction complete! duration_ms=%lld", static_cast<long long>(ms));
  host_log(LogLevel::Info, "BVH Stats nodes=%d leaves=%d depth=%d", flat->node_count, flat->leaf_count,
           flat->max_depth);
  host_log(LogLevel::Info, "Preloaded textures count=%d", static_cast<int>(flat->textures.size()));
  const mirt_scene_desc d = flat->desc();
  mirt_build_options build;
  std::memset(&build, 0, sizeof build);
  build.builder = g_config.gpu_bvh ? MIRT_BUILDER_DEVICE : MIRT_BUILDER_HOST;
  if (g_config.devices > 1) {
    // the frame's tiles are dealt to `devices` GPUs, each with its own replica of the scene
    std::vector<int> devs;
    // (MIRT_DEVICES_SAME=1 puts every member on `device`: a rehearsal of the several-GPU path on one GPU)
    const bool same = std::getenv("MIRT_DEVICES_SAME") != nullptr;
    for (int i = 0; i < g_config.devices; ++i) devs.push_back(g_config.device + (same ? 0 : i));
    if (mirt_group_create_with(&d, devs.data(), g_config.devices, &build, &device_group) != MIRT_OK)
      throw std::runtime_error(std::string("mirt_group_create: ") + mirt_last_error());
    host_log(LogLevel::Info, "Rendering on %d GPUs, frame gathered by %s", g_config.devices,
             mirt_group_transport(device_group));
  } else if (mirt_scene_create_with(&d, g_config.device, &build, &device_scene) != MIRT_OK) {
    throw std::runtime_error(std::string("mirt_scene_create: ") + mirt_last_error());
  }
}

Raytracer::~Raytracer() {
  mirt_scene_destroy(device_scene);
  mirt_group_destroy(device_group);
}

const mirt_stats &Raytracer::last_stats() const { return *stats; }

Image Raytracer::render(const Camera &camera, int width, int height, int samples_per_pixel) {
  Image output(height, width, 255, "P3");
  host_log(LogLevel::Info, "Rendering started width=%d height=%d samples=%d", width, height, samples_per_pixel);
  const mirt_camera cam = flatten_camera(camera);
  const mirt_config cfg = flatten_config(g_config);
  std::vector<double> rgb(static_cast<std::size_t>(width) * height * 3);
  // Like the reference, row times go to Output/render_row_times.csv when that directory exists in the
  // working directory (Raytracer.cpp:518-521); here: when the last primary sample of the row's band
  // of 16 rows was shaded, in milliseconds since the frame started (single-GPU renders only).
  std::FILE *csv = device_scene ? std::fopen("Output/render_row_times.csv", "w") : nullptr;
  if (device_scene) mirt_scene_record_progress(device_scene, csv != nullptr);
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = device_group
                     ? mirt_group_render(device_group, &cam, &cfg, width, height, samples_per_pixel, rgb.data(),
                                         stats.get(), nullptr)
                     : mirt_render(device_scene, &cam, &cfg, width, height, samples_per_pixel, rgb.data(), stats.get());
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (csv) {
    std::vector<double> band((static_cast<std::size_t>(height) + 15) / 16, 0.0);
    if (rc == MIRT_OK) mirt_scene_progress(device_scene, band.data(), static_cast<int>(band.size()));
    std::fprintf(csv, "Row,Time(ms)\n");
    for (int y = 0; y < height; ++y) std::fprintf(csv, "%d,%.3f\n", y, band[static_cast<std::size_t>(y) / 16]);
    std::fclose(csv);
  }
  if (rc != MIRT_OK) throw std::runtime_error(std::string("mirt_render: ") + mirt_last_error());
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const double *p = &rgb[(static_cast<std::size_t>(y) * width + x) * 3];
      output.pixels[y][x] = Pixel(p[0], p[1], p[2]);
    }
  const double rays = static_cast<double>(stats->primary_rays + stats->secondary_rays);
  host_log(LogLevel::Info, "=== Render Statistics ===");
  host_log(LogLevel::Info, "Render time seconds=%.3f", sec);
  host_log(LogLevel::Info, "Device time ms=%.3f launches=%llu", stats->device_ms,
           static_cast<unsigned long long>(stats->launches));
  host_log(LogLevel::Info, "Rays primary=%llu secondary=%llu shadow_segments=%llu Mrays/s=%.1f",
           static_cast<unsigned long long>(stats->primary_rays), static_cast<unsigned long long>(stats->secondary_rays),
           static_cast<unsigned long long>(stats->shadow_segments),
           stats->device_ms > 0 ? rays / stats->device_ms * 1e-3 : 0.0);
  host_log(LogLevel::Info, "Average time per row ms_per_row=%.4f", sec * 1e3 / (height > 0 ? height : 1));
  return output;
}

// shadow_samples is accepted and ignored, as in the reference: the engine reads
// g_config.shadow_samples (Raytracer.cpp:612-621).
Image render_scene_bvh_antialiased(const Scene &scene, const Camera &camera, int width, int height,
                                   int samples_per_pixel, int /*shadow_samples*/) {
  Raytracer rt(scene);
  return rt.render(camera, width, height, samples_per_pixel);
}
