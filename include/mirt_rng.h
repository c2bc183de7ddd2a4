/* mirt_rng.h -- the counter-based random number function shared by the HIP engine and
 * by the CPU checker, so that both draw identical numbers for identical (pixel, sample,
 * ray-tree node, dimension) coordinates.  Integer arithmetic only: results are bit-exact
 * on every target.
 *
 * The reference renderer draws from two thread-local std::mt19937 generators seeded from
 * std::random_device (Code/Render/Raytracer.cpp:75-80, Code/Math/Random.h:6-10), consumed
 * in depth-first order; that order cannot be reproduced by a breadth-first (wavefront)
 * tracer, so every draw here is addressed instead of sequenced:
 *
 *   sample key   K0 = mirt_rng_sample_key(seed, pixel, sample)
 *   child key    Kc = mirt_rng_child_key(Kparent, slot)     (slot: see MIRT_SLOT_*)
 *   a draw       u  = mirt_rng_draw(K, dim)  in [0,1), 53 bits (dim: see MIRT_DIM_*)
 */
#ifndef MIRT_RNG_H
#define MIRT_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MIRT_HD __host__ __device__
#else
#define MIRT_HD
#endif

/* dimensions of the node that owns key K */
#define MIRT_DIM_JITTER_U 0u /* Raytracer.cpp:533 */
#define MIRT_DIM_JITTER_V 1u /* Raytracer.cpp:534 */
#define MIRT_DIM_TIME 2u     /* Raytracer.cpp:538 */
#define MIRT_DIM_LENS_R1 3u  /* Camera.cpp:95 */
#define MIRT_DIM_LENS_R2 4u  /* Camera.cpp:96 */
/* shadow stratum j (running count over lights, then strata i-major / j-minor,
 * Raytracer.cpp:160-163): dims MIRT_DIM_SHADOW + 2j and + 2j + 1 */
#define MIRT_DIM_SHADOW 8u
/* glossy lobe sample i (Raytracer.cpp:360-361): dims MIRT_DIM_GLOSSY + 2i and + 2i + 1 */
#define MIRT_DIM_GLOSSY 0x100000u

/* child slots */
#define MIRT_SLOT_GLOSSY 0u          /* + lobe sample index; the mirror ray uses index 0 */
#define MIRT_SLOT_GLASS_REFLECT 0x100000u
#define MIRT_SLOT_GLASS_REFRACT 0x100001u

static inline MIRT_HD uint64_t mirt_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

static inline MIRT_HD uint64_t mirt_rng_sample_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
  uint64_t k = mirt_mix64(seed + 0x9E3779B97F4A7C15ULL * ((uint64_t)pixel + 1u));
  return mirt_mix64(k + 0xD1B54A32D192ED03ULL * ((uint64_t)sample + 1u));
}

static inline MIRT_HD uint64_t mirt_rng_child_key(uint64_t key, uint32_t slot) {
  return mirt_mix64(key + 0x8CB92BA72F3D8DD7ULL * ((uint64_t)slot + 1u));
}

static inline MIRT_HD uint64_t mirt_rng_bits(uint64_t key, uint32_t dim) {
  return mirt_mix64(key + 0x9E3779B97F4A7C15ULL * ((uint64_t)dim + 1u));
}

static inline MIRT_HD double mirt_rng_draw(uint64_t key, uint32_t dim) {
  return (double)(mirt_rng_bits(key, dim) >> 11) * (1.0 / 9007199254740992.0);
}

#endif /* MIRT_RNG_H */
