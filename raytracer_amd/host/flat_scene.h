// The flat, engine-ready form of a scene: the plain arrays of include/mirt.h with their storage.
// No dependence on any scene class, so that it can be filled from the drop-in classes
// (mirt_scene.h) and from the reference's own (Code/Core/Scene.h) alike: see flatten_impl.h.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "mirt.h"

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

inline mirt_scene_desc FlatScene::desc() const {
  mirt_scene_desc d;
  std::memset(&d, 0, sizeof d);
  d.n_nodes = static_cast<int32_t>(nodes.size());
  d.n_leaf_prims = static_cast<int32_t>(leaf_prims.size());
  d.n_prims = static_cast<int32_t>(prims.size());
  d.n_plane_points = static_cast<int32_t>(plane_points.size() / 3);
  d.n_materials = static_cast<int32_t>(materials.size());
  d.n_lights = static_cast<int32_t>(lights.size());
  d.n_textures = static_cast<int32_t>(textures.size());
  d.nodes = nodes.data();
  d.leaf_prims = leaf_prims.data();
  d.prims = prims.data();
  d.plane_points = plane_points.data();
  d.materials = materials.data();
  d.lights = lights.data();
  d.textures = textures.data();
  std::memcpy(d.background_color, background_color, sizeof background_color);
  d.background_strength = background_strength;
  return d;
}
