// Scene -> mirt_scene_desc: the host half of `Raytracer::Raytracer` (Code/Render/Raytracer.cpp:26-71):
// global primitive numbering, BVH build with the reference's split rule (Code/Geometry/Bvh.cpp:16-242),
// texture preload (Raytracer.cpp:82-110) and conversion to the plain arrays of include/mirt.h.
#pragma once

#include <string>
#include <vector>

#include "mirt.h"
#include "mirt_scene.h"

struct FlatScene {
  std::vector<mirt_node> nodes;        // depth-first order, root first
  std::vector<int32_t> leaf_prims;     // primitive indices, leaf by leaf
  std::vector<mirt_prim> prims;        // global order: spheres, cubes, planes, toruses, cylinders, cones
  std::vector<double> plane_points;
  std::vector<mirt_material> materials;
  std::vector<mirt_light> lights;
  std::vector<mirt_texture> textures;
  std::vector<std::string> texture_names;
  std::vector<std::vector<uint16_t>> texels;  // storage behind textures[i].rgb
  double background_color[3] = {0, 0, 0};
  double background_strength = 1.0;

  int node_count = 0, leaf_count = 0, max_depth = 0;  // BVHStats (Bvh.h:62-66)

  mirt_scene_desc desc() const;
};

// texture_root: directory that holds `Textures/`; empty = the current directory, as in the
// reference (Code/IO/Texture.cpp:37-45).
// build_tree = false leaves nodes and leaf_prims empty, for an engine scene whose tree is built on the
// device (mirt_scene_create_with, MIRT_BUILDER_DEVICE).
FlatScene flatten_scene(const Scene &scene, const std::string &texture_root = "", bool build_tree = true);

mirt_camera flatten_camera(const Camera &camera);
mirt_config flatten_config(const RenderConfig &config);
