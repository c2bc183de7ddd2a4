// TEST INFRASTRUCTURE ONLY -- never included by product code.
//
// Force-included (g++ -include) in front of the two reference translation units that
// own a random generator (Code/Render/Raytracer.cpp:75-80 and Code/Math/Random.h:6-10,
// the latter pulled in by Code/Core/Camera.cpp) when they are compiled for the golden
// harness.  The reference seeds `std::mt19937` from std::random_device, so two runs of
// it never agree; this header swaps the engine type for one that hands out 32-bit words
// chosen by the harness, leaving every line of the reference's own arithmetic as it is.
//
// std::uniform_real_distribution<double> on a 32-bit engine draws two words g1, g2 and
// returns (g1 + g2 * 2^32) / 2^64.  The harness splits a 53-bit value k (u = k / 2^53)
// as X = k << 11, g1 = low32(X), g2 = high32(X), so the reference's random_double()
// returns exactly u.
#ifndef ORACLE_REF_RNG_HOOK_H
#define ORACLE_REF_RNG_HOOK_H

#include <cstdint>
#include <random>

extern "C" std::uint32_t oracle_hook_next_word();

namespace std {
struct oracle_hook_engine {
  using result_type = std::uint32_t;
  template <class T> explicit oracle_hook_engine(T) {}
  oracle_hook_engine() {}
  static constexpr result_type min() { return 0u; }
  static constexpr result_type max() { return 0xffffffffu; }
  result_type operator()() { return oracle_hook_next_word(); }
};
}  // namespace std

#define mt19937 oracle_hook_engine

#endif
