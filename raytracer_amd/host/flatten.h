// Scene -> mirt_scene_desc: the host half of `Raytracer::Raytracer` (Code/Render/Raytracer.cpp:26-71):
// global primitive numbering, BVH build with the reference's split rule (Code/Geometry/Bvh.cpp:16-242),
// texture preload (Raytracer.cpp:82-110) and conversion to the plain arrays of include/mirt.h.
// The work is done by the templates of flatten_impl.h; these are their instances for the drop-in classes.
#pragma once

#include <string>
#include <vector>

#include "flat_scene.h"
#include "mirt.h"
#include "mirt_scene.h"

// texture_root: directory that holds `Textures/`; empty = the current directory, as in the
// reference (Code/IO/Texture.cpp:37-45).
// build_tree = false leaves nodes and leaf_prims empty, for an engine scene whose tree is built on the
// device (mirt_scene_create_with, MIRT_BUILDER_DEVICE).
FlatScene flatten_scene(const Scene &scene, const std::string &texture_root = "", bool build_tree = true);

mirt_camera flatten_camera(const Camera &camera);
mirt_config flatten_config(const RenderConfig &config);
