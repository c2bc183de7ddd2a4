// Synthetic large scenes (no counterpart in the reference): a cloud of small spheres, cylinders,
// cones and tori, built directly as a Scene so that it takes the same flattening and BVH build as a
// parsed file without a gigabyte of scene text in between.
#pragma once

#include <cstdint>
#include <string>

#include "mirt_scene.h"

// n_prims primitives, a quarter of each kind (the remainder goes to the spheres), centres uniform in
// [-100, 100]^3, sizes log-uniform in [0.05, 0.5], uniform Euler angles, no motion; 20 % mirrors
// (reflectivity 0.9, glossiness 1), 5 % glass (transparency 1, refractive index 1.45), the rest
// diffuse; three point lights; one 16:9 camera outside the cloud looking at its centre.
// The same (n_prims, seed) always gives the same scene.
Scene make_synthetic_scene(int n_prims, std::uint64_t seed);

// Writes `scene` in the ASCII grammar load_scene reads (spheres, cylinders, cones, tori, point
// lights and cameras only).  Returns false when the file cannot be written.
bool write_scene_ascii(const Scene &scene, const std::string &path);
