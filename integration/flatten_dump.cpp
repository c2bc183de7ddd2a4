// Test driver (built by oracle/Makefile against the reference's headers and objects): loads a scene
// with the reference's own load_scene, flattens it with the templates of flatten_impl.h instantiated
// for the reference's classes, and writes every array of the result to a file, so that
// tests/test_reference_binding.py can compare it byte for byte with what the drop-in host library
// makes of the same file.
#include <cstdio>
#include <string>

#include "Core/Config.h"
#include "Core/Scene.h"
#include "IO/Image.h"
#include "IO/SceneUtils.h"

#include "flatten_impl.h"

namespace {
template <class T> void put(std::FILE *f, const char *name, const T *data, std::size_t n) {
  const unsigned long long bytes = n * sizeof(T);
  std::fprintf(f, "%s %llu\n", name, bytes);
  if (bytes) std::fwrite(data, 1, bytes, f);
  std::fputc('\n', f);
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: flatten_dump <scene.txt> <texture root> <out>\n");
    return 2;
  }
  const Scene scene = load_scene(argv[1]);
  const FlatScene flat = mirt_flatten::scene<Image>(scene, argv[2]);
  std::FILE *f = std::fopen(argv[3], "wb");
  if (!f) return 1;
  put(f, "nodes", flat.nodes.data(), flat.nodes.size());
  put(f, "leaf_prims", flat.leaf_prims.data(), flat.leaf_prims.size());
  put(f, "prims", flat.prims.data(), flat.prims.size());
  put(f, "plane_points", flat.plane_points.data(), flat.plane_points.size());
  put(f, "materials", flat.materials.data(), flat.materials.size());
  put(f, "lights", flat.lights.data(), flat.lights.size());
  for (std::size_t i = 0; i < flat.textures.size(); ++i) {
    const int dims[3] = {flat.textures[i].width, flat.textures[i].height, flat.textures[i].max_value};
    put(f, "texture_dims", dims, 3);
    put(f, "texels", flat.texels[i].data(), flat.texels[i].size());
  }
  const double bg[4] = {flat.background_color[0], flat.background_color[1], flat.background_color[2], flat.background_strength};
  put(f, "background", bg, 4);
  if (!scene.cameras.empty()) {
    const mirt_camera cam = mirt_flatten::camera(scene.cameras[0]);
    put(f, "camera", &cam, 1);
  }
  const mirt_config cfg = mirt_flatten::config(g_config);  // RenderConfig's defaults (Config.h)
  put(f, "config", &cfg, 1);
  std::fclose(f);
  return 0;
}
