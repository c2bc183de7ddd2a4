// The binding a maintainer of the reference adds to put the MI355X engine behind the reference's own
// classes: a replacement for the one function through which the render path is entered,
//     Image render_scene_bvh_antialiased(const Scene&, const Camera&, int w, int h, int spp, int)
//     (Code/Render/Raytracer.h:65-67, defined at Code/Render/Raytracer.cpp:612-621, called from
//     Code/Core/Main.cpp:196-198).
// Compile this file INSTEAD OF Code/Render/Raytracer.cpp, with the reference's include paths plus
// -I<repo>/include -I<repo>/raytracer_amd/host, and link -lmirt; everything else of the reference -- main(),
// the flag parser, load_scene, Scene / Camera / Image, the PPM writer -- stays as it is.  The scene is
// flattened by the same templates the drop-in host library uses (flatten_impl.h), instantiated here for
// the reference's classes.  oracle/Makefile builds the reference's command line this way
// (oracle/_ref/raytracer_ref_gpu); tests/test_reference_binding.py checks the flattening byte for byte
// and, on a GPU, the image.
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "Core/Config.h"
#include "Render/Raytracer.h"

#include "flatten_impl.h"
#include "mirt.h"

Image render_scene_bvh_antialiased(const Scene &scene, const Camera &camera, int width, int height,
                                   int samples_per_pixel, int /*shadow_samples: ignored, as at Raytracer.cpp:612-621*/) {
  // Raytracer::Raytracer (Raytracer.cpp:26-71): number the objects, build the BVH, preload the textures
  const FlatScene flat = mirt_flatten::scene<Image>(scene);
  const mirt_scene_desc desc = flat.desc();
  const char *dev = std::getenv("MIRT_DEVICE");
  mirt_scene *gpu = nullptr;
  if (mirt_scene_create(&desc, dev ? std::atoi(dev) : 0, &gpu) != MIRT_OK)
    throw std::runtime_error(std::string("mirt_scene_create: ") + mirt_last_error());

  // Raytracer::render (Raytracer.cpp:475-607) with the g_config fields it reads
  const mirt_camera cam = mirt_flatten::camera(camera);
  const mirt_config cfg = mirt_flatten::config(g_config);
  std::vector<double> rgb(static_cast<std::size_t>(width) * height * 3);
  mirt_stats stats;
  const int rc = mirt_render(gpu, &cam, &cfg, width, height, samples_per_pixel, rgb.data(), &stats);
  const std::string why = rc == MIRT_OK ? "" : mirt_last_error();
  mirt_scene_destroy(gpu);
  if (rc != MIRT_OK) throw std::runtime_error("mirt_render: " + why);

  Image out(height, width, 255, "P3");  // Raytracer.cpp:476
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const double *p = &rgb[(static_cast<std::size_t>(y) * width + x) * 3];
      out.pixels[y][x] = Pixel(p[0], p[1], p[2]);  // post-gamma, un-clamped, as Raytracer.cpp:570 stores them
    }
  return out;
}
