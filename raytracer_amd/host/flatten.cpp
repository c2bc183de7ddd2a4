// Scene -> flat arrays for the drop-in classes: instances of the templates of flatten_impl.h.
#include "flatten.h"

#include "flatten_impl.h"

FlatScene flatten_scene(const Scene &scene, const std::string &texture_root, bool build_tree) {
  return mirt_flatten::scene<Image>(scene, texture_root, build_tree);
}

mirt_camera flatten_camera(const Camera &c) { return mirt_flatten::camera(c); }

mirt_config flatten_config(const RenderConfig &c) { return mirt_flatten::config(c); }
