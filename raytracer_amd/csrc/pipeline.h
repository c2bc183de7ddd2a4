// The render pipeline kernels (gfx950, 64-lane waves):
//
//   k_trace<PRIMARY>   one lane per ray: generate it (PRIMARY) or read it from the depth's ray queue,
//                      find the closest hit by one of the BVH walks below, shade it without
//                      visibility (ambient, emission, per-light Blinn-Phong terms), append one shadow
//                      work item per contributing light and the reflection / glossy / refraction rays
//                      to the next depth's queue (wave ballot + one atomic per wave on a sharded tail).
//   k_shadow           one lane per (hit, light): walks the strata of the light, each as a chain of
//                      closest-hit segments through transparent objects, then adds the light's
//                      contribution times its visibility to the pixel.
//   k_resolve          mean over samples, tone map, gamma; leaves the accumulator zeroed.
//   k_trace_ids, k_trace_rays   parity hooks: closest hit of pixel-centre rays / of given rays.
//
// The ray tree of the reference's recursive trace() (Code/Render/Raytracer.cpp:296-473) is linear in
// the colours its children return, so every ray carries the RGB weight with which its radiance
// enters the pixel and the tree is processed one depth at a time.  Launches that consume a queue
// read its length from the producer's tail counters on the device; the host never waits inside a frame.
//
// BVH walks: nodes are stored in the reference's own visiting order with skip links, so the
// reference's walk needs no stack.  traverse_packet drives it per WAVE: the node index is
// wave-uniform, node and primitive records are fetched once per wave through the scalar data path,
// and every lane tests its own ray against them (lanes whose ray missed a box sleep until the walk
// leaves that subtree); primitive kind and transform kind are then uniform branches, so the six
// intersectors never diverge against each other inside a wave.  traverse_lane walks per lane in the
// same order (large scenes); traverse_ordered walks per lane, nearer child first (opt-in).
#pragma once

#include "geometry.h"
#include "mirt_rng.h"

#define WAVE 64

// Register budget of the two big kernels, as waves per SIMD the compiler must make room for.  The
// packet walk is bound by instruction issue, one wave's scalar and vector instructions in turn: a fourth
// wave is worth 3 % (k_trace) to 10 % (k_shadow with strata) once the kernels fit 128 registers without
// spilling, which they do since loop-invariant constants are no longer hoisted (see the Makefile) and the
// kernel arguments are read where they are needed (args_again); the variants that carry texture
// coordinates, and the shadow walk that chains segments through attenuating surfaces, need more and stay
// at three.  The per-lane walks wait on dependent node fetches most of the time: there a fourth wave is
// worth more than the registers it takes away (measured 12 % on a million primitives).
#ifndef MIRT_TRACE_WAVES
#define MIRT_TRACE_WAVES 4
#endif
#ifndef MIRT_SHADOW_WAVES
#define MIRT_SHADOW_WAVES 4
#endif
#ifndef MIRT_LANE_WALK_WAVES
#define MIRT_LANE_WALK_WAVES 4
#endif
#define WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#ifndef MIRT_WIDE_WALK_WAVES
#define MIRT_WIDE_WALK_WAVES 3
#endif
#define WAVES_FOR_WALK(walk, packet_waves)                                                   \
  WAVES_PER_SIMD((walk) == WALK_PACKET ? (packet_waves) : (walk) == WALK_ORDERED ? 3 : (walk) == WALK_WIDE \
                     ? MIRT_WIDE_WALK_WAVES : MIRT_LANE_WALK_WAVES)
// Keeps the compiler from carrying values in registers across a section whose register budget is
// tight: what is stored to LDS before the fence has to be loaded again after the next one.
#define PARK_FENCE() asm volatile("" ::: "memory")

// Development aid: -DMIRT_PROFILE=1 (trace kernel) or =2 (shadow kernel) replaces the statistics
// counters by shader-clock cycles per wave spent in the kernel's sections.
#ifdef MIRT_PROFILE
#define PROF_DECL unsigned long long prof_t[6] = {0, 0, 0, 0, 0, 0}, prof_last = clock64()
#define PROF_MARK(i)                         \
  do {                                       \
    const unsigned long long now_ = clock64(); \
    prof_t[i] += now_ - prof_last;           \
    prof_last = now_;                        \
  } while (0)
#define PROF_FLUSH(c)                                                       \
  do {                                                                      \
    if (__lane_id() == 0) {                                                 \
      atomicAdd(&(c)->node_tests, prof_t[0]);                               \
      atomicAdd(&(c)->prim_tests, prof_t[1]);                               \
      atomicAdd(&(c)->shadow_node_tests, prof_t[2]);                        \
      atomicAdd(&(c)->shadow_prim_tests, prof_t[3]);                        \
      atomicAdd(&(c)->shadow_segments, prof_t[4]);                          \
    }                                                                       \
  } while (0)
#else
#define PROF_DECL
#define PROF_MARK(i)
#define PROF_FLUSH(c)
#endif

DEV double maxd(double a, double b) { return (a < b) ? b : a; }  // std::max(a, b)
DEV double mind(double a, double b) { return (b < a) ? b : a; }  // std::min(a, b)

// 1/sqrt(x) and 1/x for the colour terms only: the hardware estimate and two Newton steps (relative
// error of the order of 1e-16, not correctly rounded) instead of the IEEE sqrt and division sequences,
// which cost three to five times the instructions.  x > 0 and finite.
DEV double rsqrt_shade(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}
DEV double rcp_shade(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}

// x^y for the colour terms only (specular lobe): log2 by the atanh series on the mantissa folded into
// [sqrt(1/2), sqrt(2)), exp2 by the Taylor series on the fraction.  Relative error below 1e-13 for the
// exponents materials use (y up to a few thousand), at well under half the instructions of the library
// pow, which is correctly rounded.  Nothing on the geometric chain (hit points, normals, ray
// directions) goes through this: those keep the library functions.
DEV double pow_shade(double x, double y) {
  if (!(x > 0.0)) return (y == 0.0) ? 1.0 : 0.0;
  int e;
  double mant = frexp(x, &e);  // [1/2, 1)
  if (mant < 0.70710678118654752440) {
    mant *= 2.0;
    e -= 1;
  }
  const double z = (mant - 1.0) * rcp_shade(mant + 1.0), z2 = z * z;  // |z| <= 0.1716
  double p = 1.0 / 23.0;
  p = p * z2 + 1.0 / 21.0;
  p = p * z2 + 1.0 / 19.0;
  p = p * z2 + 1.0 / 17.0;
  p = p * z2 + 1.0 / 15.0;
  p = p * z2 + 1.0 / 13.0;
  p = p * z2 + 1.0 / 11.0;
  p = p * z2 + 1.0 / 9.0;
  p = p * z2 + 1.0 / 7.0;
  p = p * z2 + 1.0 / 5.0;
  p = p * z2 + 1.0 / 3.0;
  p = p * z2 + 1.0;
  const double lg = (2.0 * z * p) * 1.44269504088896340736 + (double)e;  // log2(x)
  const double t = y * lg;
  if (t < -1070.0) return 0.0;
  if (t > 1020.0) return __builtin_huge_val();
  const double n = rint(t);
  const double f = (t - n) * 0.69314718055994530942;  // |f| <= 0.3466
  double q = 1.0 / 6227020800.0;
  q = q * f + 1.0 / 479001600.0;
  q = q * f + 1.0 / 39916800.0;
  q = q * f + 1.0 / 3628800.0;
  q = q * f + 1.0 / 362880.0;
  q = q * f + 1.0 / 40320.0;
  q = q * f + 1.0 / 5040.0;
  q = q * f + 1.0 / 720.0;
  q = q * f + 1.0 / 120.0;
  q = q * f + 1.0 / 24.0;
  q = q * f + 1.0 / 6.0;
  q = q * f + 0.5;
  q = q * f + 1.0;
  q = q * f + 1.0;
  return ldexp(q, (int)n);
}

// A node as the per-lane walks read it: the whole 64-byte record in four 16-byte loads issued
// together, and a box test without early exits, so that nothing of the record is fetched behind a
// branch.  (Read field by field inside the early-exit test, a node costs four to five dependent
// round trips to memory per step -- x slabs, y slabs, z slabs, then the links -- which is what the
// per-lane walks were bound by: about 3000 cycles a step on a million primitives.)
struct NodeVal {
  double b[6];  // bmin, bmax
  int32_t skip, first, count, order;
};
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int16v __attribute__((ext_vector_type(16)));
typedef float flt4v __attribute__((ext_vector_type(4)));
DEV NodeVal load_node(const DNode *p) {
  const dbl2 *q = reinterpret_cast<const dbl2 *>(p);
  const dbl2 a = q[0], b = q[1], c = q[2];
  const int4v m = reinterpret_cast<const int4v *>(p)[3];
  NodeVal n;
  n.b[0] = a.x; n.b[1] = a.y; n.b[2] = b.x; n.b[3] = b.y; n.b[4] = c.x; n.b[5] = c.y;
  n.skip = m.x; n.first = m.y; n.count = m.z; n.order = m.w;
  return n;
}
// The same without branches, for the per-lane walks.  The reference returns at the first axis that fails; carrying on and remembering the failure gives the same
// answer (the bounds up to the first failure are the same).
DEV bool box_hit_v(const double *lo, const double *hi, const RayL &r, double t_min, double t_max) {
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double inv = comp(r.inv, a), o = comp(r.o, a);
    double t0 = (lo[a] - o) * inv;
    double t1 = (hi[a] - o) * inv;
    const bool neg = inv < 0.0;
    const double n0 = neg ? t1 : t0, n1 = neg ? t0 : t1;
    t_min = maxd(n0, t_min);
    t_max = mind(n1, t_max);
    ok = ok & !(t_max <= t_min);
  }
  return ok;
}

// BoundingBox::intersect (BoundingBox.h:14-31) as written, NaN behaviour of std::max / std::min and the
// early exits included, for values in scalar registers (packet walk).
DEV bool box_hit_early(const double *lo, const double *hi, const RayL &r, double t_min, double t_max) {
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double inv = comp(r.inv, a), o = comp(r.o, a);
    double t0 = (lo[a] - o) * inv;
    double t1 = (hi[a] - o) * inv;
    if (inv < 0.0) {
      const double s = t0;
      t0 = t1;
      t1 = s;
    }
    t_min = maxd(t0, t_min);
    t_max = mind(t1, t_max);
    if (t_max <= t_min) return false;
  }
  return true;
}

// intersect_bvh (Bvh.cpp:306-333) for the 64 rays of a wave at once.  Must be called by all lanes
// of the wave; `active` masks the lanes that have a ray.
// ANYHIT (shadow segments of a scene whose primitives are all opaque or perfectly clear): clear primitives
// are not tested, and a lane that has found a hit nearer than `any_limit` takes no further part.
template <bool SHADOW, bool UV, bool MOTION, bool ANYHIT = false>
DEV void traverse_packet(const DScene &sc, const RayL &ray, bool active, double t_min, double &closest_t,
                         Best<UV> &best, unsigned &n_node, unsigned &n_prim, double any_limit = 0.0) {
  const int n_nodes = sc.n_nodes;
  int resume = active ? 0 : 0x7fffffff;  // first node at which this lane takes part again
  int node = 0;
  while (node < n_nodes) {
    const int un = __builtin_amdgcn_readfirstlane(node);
    // the whole record in one scalar load: read field by field inside the box test it cost six
    // dependent trips to the scalar cache per node (x, y, z slabs, then each link)
    int16v raw = *reinterpret_cast<const RO int16v *>(rdonly(sc.nodes) + un);
    asm volatile("" : "+s"(raw));
    const NodeVal nd = __builtin_bit_cast(NodeVal, raw);
    const bool on = resume <= un;
    // Every lane runs the whole test (BoundingBox.h:14-31 without its early returns, same answer): a wave
    // pays for all three axes as soon as one lane passes the first two, and the three exits cost more
    // scalar instructions than they save vector ones (+2.9 % on Test5 1080p x 16).
    bool hit = on & box_hit_v(nd.b, nd.b + 3, ray, t_min, closest_t);
    n_node += on ? 1u : 0u;
    resume = (on & !hit) ? nd.skip : resume;
    if (!__any(hit)) {
      node = nd.skip;
      continue;
    }
    if (nd.first >= 0) {
      for (int k = 0; k < nd.count; k++) {
        // (the first primitive's entry came with the node record: one dependent scalar load less per leaf; +1 % on Test5)
        const int tag = k == 0 ? nd.order : rdonly(sc.leaf_tagged)[nd.first + k];
        if (ANYHIT && (tag & LEAF_CLEAR_BIT)) continue;
        const int pi = tag & LEAF_PRIM_MASK;
        if (hit) {
          n_prim++;
          isect_prim<SHADOW, UV, MOTION>(sc, rdonly(sc.prims) + pi, pi, ray, t_min, closest_t, best);
          if (ANYHIT && best.prim >= 0 && best.t < any_limit) {
            hit = false;
            resume = 0x7fffffff;
          }
        }
      }
      node = nd.skip;
    } else {
      node = un + 1;
    }
  }
}

// The same walk with one node index per LANE, for scenes whose rays do not share their path through
// the tree: each lane follows its own skip links until it stands on a leaf whose box it hits (or
// runs out of nodes), then all lanes intersect their leaf's primitives together.  Node and primitive
// records are fetched per lane; the branches on primitive and transform kind diverge, so a wave
// pays for every kind some lane holds.  The sequence of tests of each ray is the same as in the
// packet walk, and so is the result.
#ifndef LANE_FEW_WALKERS
#define LANE_FEW_WALKERS 24  // stop walking for now when at most this many lanes still walk ...
#endif
#ifndef LANE_MANY_PENDING
#define LANE_MANY_PENDING 8  // ... and at least this many hold a leaf to intersect
#endif
template <bool SHADOW, bool UV, bool MOTION>
DEV void traverse_lane(const DScene &sc, const RayL &ray, bool active, double t_min, double &closest_t,
                       Best<UV> &best, unsigned &n_node, unsigned &n_prim) {
  const int n_nodes = sc.n_nodes;
  int node = active ? 0 : n_nodes;
  int first = 0, count = 0;  // the leaf this lane holds, to be intersected at the next drain
  while (__any(node < n_nodes || count > 0)) {
    // Walk: lanes without a leaf follow their skip links.  The phase ends when nobody walks any more,
    // or early when only a few stragglers do while many lanes wait with a leaf: a walking step
    // costs the wave the same with 2 lanes as with 60, and the stragglers simply go on next round
    // (a lane's own sequence of tests does not depend on when it waits).
    for (;;) {
      const bool walking = node < n_nodes && count == 0;
      const int n_walking = __popcll(__ballot(walking));
      if (n_walking == 0) break;
      if (n_walking <= LANE_FEW_WALKERS && __popcll(__ballot(count > 0)) >= LANE_MANY_PENDING) break;
      if (walking) {
        const NodeVal nd = load_node(sc.nodes + node);
        n_node++;
        const bool hit = box_hit_v(nd.b, nd.b + 3, ray, t_min, closest_t);
        const bool leaf = nd.first >= 0;
        if (hit && leaf) {
          first = nd.first;
          count = nd.count;
        }
        node = (hit && !leaf) ? node + 1 : nd.skip;
      }
    }
    // Drain: every lane that holds a leaf intersects its primitives, in leaf order.
    for (int k = 0; __any(k < count); k++) {
      if (k < count) {
        const int pi = rdonly(sc.leaf_prims)[first + k];
        n_prim++;
        isect_prim_lane<SHADOW, UV, MOTION>(sc, sc.prims + pi, pi, ray, t_min, closest_t, best);
      }
    }
    count = 0;
  }
}

// Per-lane walk that descends into the nearer child first and keeps the farther one on a per-lane
// stack in LDS: the closest hit is found early and most of the far subtrees are then culled by their
// box test against it.  The reference's walk has no such order (Bvh.cpp:322-332), which on a deep
// tree costs it several times as many box tests for the same answer.  "Nearer" is decided by the sign
// of the ray direction along the axis on which the two children lie furthest apart.  The closest
// hit does not depend on the visiting order except when two surfaces are hit at exactly the same
// distance, where the one tested later can win (each intersector has its own rule for equality).
template <bool SHADOW, bool UV, bool MOTION>
DEV void traverse_ordered(const DScene &sc, const RayL &ray, bool active, double t_min, double &closest_t,
                          Best<UV> &best, unsigned &n_node, unsigned &n_prim) {
  __shared__ int32_t stack[WALK_STACK][256];
  const int tid = threadIdx.x;
  int sp = 0;
  int node = active ? 0 : -1;
  while (__any(node >= 0)) {
    int first = 0, count = 0;
    while (node >= 0 && count == 0) {
      const NodeVal nd = load_node(sc.nodes + node);
      n_node++;
      int next = -1;
      if (box_hit_v(nd.b, nd.b + 3, ray, t_min, closest_t)) {
        if (nd.first >= 0) {
          first = nd.first;
          count = nd.count;
        } else {
          const int axis = nd.order & 3;
          const double d = comp(ray.d, axis);
          // second child first when it is the lower one and the ray runs upwards, or the reverse
          const bool second_first = ((nd.order & 4) != 0) == (d >= 0.0);
          next = second_first ? nd.count : node + 1;
          stack[sp++][tid] = second_first ? node + 1 : nd.count;
        }
      }
      if (next < 0 && sp > 0) next = stack[--sp][tid];
      node = next;
    }
    for (int k = 0; __any(k < count); k++) {
      if (k < count) {
        const int pi = rdonly(sc.leaf_prims)[first + k];
        n_prim++;
        isect_prim_lane<SHADOW, UV, MOTION>(sc, sc.prims + pi, pi, ray, t_min, closest_t, best);
      }
    }
  }
}

// The throughput walk for large scenes.  Measured on a million primitives, the walks above spend
// three quarters of their time walking, about 3000 cycles per step: every node test is a dependent
// fetch from a 128 MB node array, and 23 % of the lanes are busy in a step (the others hold a leaf and
// wait, or have finished).  This walk goes through the SAME tree, collapsed two levels at a time into
// 4-wide nodes (one 128-byte line: the four grandchildren's boxes rounded outwards to float, so a
// fetch brings four tests and half the levels disappear), visits the children nearer-first in the
// order the binary ordered walk would, and batches the primitive tests by kind:
//  * a collapsed inner node's own box is never tested.  It contains its children's boxes, so a ray that
//    enters a child's box also enters it: leaving it out changes no result.  The float boxes are
//    supersets tested in double (rounding is monotone, so a superset box passes whenever the exact one
//    does); a leaf's exact double box is then tested as the reference tests it, including the rule
//    that a box of zero thickness is never hit (BoundingBox.h:27).  (Tried: 256-byte nodes with exact
//    boxes and the leaves' primitives in the slots -- no leaf visits, but twice the bytes and registers
//    per visit: 14 % slower on a million primitives);
//  * a lane that enters a leaf only notes the leaf's primitives in a small per-lane list in LDS and
//    walks on; when too few lanes can still walk, the wave picks the kind most lanes hold and every
//    lane holding one tests one primitive of that kind -- one intersector, many lanes.
// A primitive is tested against the closest hit known when its turn comes, so the closest hit is the
// same as in the other walks; boxes are tested against a closest hit that may be a little out of date.
// The order of the tests differs, which matters where two surfaces coincide (each intersector accepts on
// its own object-space distance and records the world-space one, so the later test can win by a rounding):
// closest-hit rays whose two best hits lie within the scene's tie band of each other are walked again
// in the reference's order (walk_again_in_order) when the scene has the reference's tree.
#ifndef PEND_SLOTS
#define PEND_SLOTS 8  // primitives a lane can hold untested (at most 8: one byte of slot mask per kind)
#endif
#ifndef WIDE_WALK_MIN
#define WIDE_WALK_MIN 16  // walk on while at least this many lanes can, else test a batch
#endif
#ifndef WIDE_DRAIN_MIN
#define WIDE_DRAIN_MIN 24  // test a batch as soon as this many lanes hold a primitive of one kind
#endif

#define WIDE_TIE 0x100u  // WideWalk::up: two hits of this ray lay within the scene's tie band of each other

// A lane's place in the wide walk.  The walk is written as one wave step at a time (wide_step) so that
// a kernel can interleave steps with handing finished lanes their next ray.
struct WideWalk {
  int sp;
  int32_t cur;            // >= 0: wide node; < 0: ~leaf; WIDE_NONE: nothing left to visit
  uint32_t used;          // bit s: slot s of this lane's list holds a primitive
  unsigned long long km;  // bits [8k, 8k + 8): the slots that hold a primitive of kind k
  uint32_t up;            // bits 0-2: the sign of the ray direction per axis, as the ordered walk reads it; WIDE_TIE
  DEV bool tied() const { return (up & WIDE_TIE) != 0; }
  DEV void start(const RayL &ray, bool active) {
    sp = 0;
    cur = active ? 0 : WIDE_NONE;
    used = 0;
    km = 0;
    up = (ray.d.x >= 0.0 ? 1u : 0u) | (ray.d.y >= 0.0 ? 2u : 0u) | (ray.d.z >= 0.0 ? 4u : 0u);
  }
  DEV bool busy() const { return cur != WIDE_NONE || used != 0; }
  DEV void stop() {  // drops what is left of the walk
    sp = 0;
    cur = WIDE_NONE;
    used = 0;
    km = 0;
  }
};

// One step of the wave: the lanes that can walk visit their node or leaf, or -- when too few can --
// the lanes holding a primitive of the most popular kind test it.  Returns false when no lane has
// anything left.  Must be called by all lanes of the wave.
// ANYHIT (shadow segments of a scene whose primitives are all opaque or perfectly clear): the question is
// only whether an opaque primitive lies nearer than `any_limit`, so clear primitives are not tested at all
// and a lane is done with its ray at the first hit nearer than the limit, whichever primitive it is on.
#ifndef MIRT_WIDE_PAIR
#define MIRT_WIDE_PAIR 1
#endif
template <bool SHADOW, bool UV, bool MOTION, bool ANYHIT = false>
DEV bool wide_step(const DScene &sc, const RayL &ray, double t_min, double &closest_t, Best<UV> &best, WideWalk &w,
                   int32_t *deep, unsigned &n_node, unsigned &n_prim, double any_limit = 0.0) {
  __shared__ int32_t stack_lds[WIDE_STACK_LDS][256];
  __shared__ int32_t pend[PEND_SLOTS][256];
  const int tid = threadIdx.x;
  constexpr bool PAIR = MIRT_WIDE_PAIR == 2 || (ANYHIT && MIRT_WIDE_PAIR == 1);
  auto push = [&](int32_t v) {
    if (w.sp < WIDE_STACK_LDS) stack_lds[w.sp][tid] = v;
    else deep[w.sp - WIDE_STACK_LDS] = v;
    w.sp++;
  };
  auto pop = [&]() {
    w.sp--;
    return w.sp < WIDE_STACK_LDS ? stack_lds[w.sp][tid] : deep[w.sp - WIDE_STACK_LDS];
  };
  // One primitive against this lane's ray.  Closest-hit walks keep closest_t a little beyond the closest hit
  // (by the scene's tie band, DScene::tie_hi): a surface that coincides with the closest one is then still
  // reached -- through its boxes and in its intersector -- whichever of the two the walk meets first, and
  // the lane is marked; its ray is walked again in the reference's order when this walk is over
  // (walk_again_in_order).  The old bound waits in LDS while the intersector runs, not in registers.
  __shared__ double bound_was[256];
  auto test = [&](int pi) {
    n_prim++;
    if (SHADOW) {
      isect_prim_lane<SHADOW, UV, MOTION>(sc, sc.prims + pi, pi, ray, t_min, closest_t, best);
      return;
    }
    *(volatile double *)&bound_was[tid] = closest_t;
    isect_prim_lane<SHADOW, UV, MOTION>(sc, sc.prims + pi, pi, ray, t_min, closest_t, best);
    const double was = *(volatile double *)&bound_was[tid];
    if (closest_t != was) {  // hit: closest_t is the new distance
      if (closest_t >= was * sc.tie_lo) w.up |= WIDE_TIE;
      closest_t *= sc.tie_hi;
    }
  };
  const int room = PEND_SLOTS - __popc(w.used);
  const bool walks = w.cur != WIDE_NONE && room >= 2;
  const int n_walk = __popcll(__ballot(walks));
  // the kind of primitive most lanes hold, and how many lanes hold one
  int kind = 0, most = 0;
#pragma unroll
  for (int k = 0; k <= MIRT_CONE; k++) {
    if (!((sc.kinds >> k) & 1)) continue;  // wave-uniform: the scene has no primitive of this kind
    const int c = __popcll(__ballot(((w.km >> (8 * k)) & 0xffull) != 0));
    if (c > most) {
      most = c;
      kind = k;
    }
  }
  if (n_walk == 0 && most == 0) return false;  // a lane with a node always has room once nothing is pending
  // walk, unless a full enough batch is waiting or too few lanes could take part
  if (most < WIDE_DRAIN_MIN && (n_walk >= WIDE_WALK_MIN || most == 0)) {
    if (walks) {
      const int32_t cur = w.cur;
      if (cur >= 0) {
        // A wide node: the whole 128-byte record in eight loads issued together.  PAIR (any-hit shadow
        // segments, whose rays mostly cross the whole scene unoccluded and so visit every node on their way
        // whatever the order): when the next entry on the stack is a wide node too, it is visited in the same
        // step -- both records are asked for at once, which halves the dependent fetches of the walk; its
        // children go onto the stack first, below those of the nearer node, so the order of visits is unchanged.
        int32_t second = WIDE_NONE;
        if (PAIR && w.sp > 0) {
          const int32_t top = w.sp <= WIDE_STACK_LDS ? stack_lds[w.sp - 1][tid] : deep[w.sp - 1 - WIDE_STACK_LDS];
          if (top >= 0) {
            second = top;
            w.sp--;
          }
        }
        const flt4v *wq = reinterpret_cast<const flt4v *>(sc.wide + cur);
        const flt4v f0 = wq[0], f1 = wq[1], f2 = wq[2], f3 = wq[3], f4 = wq[4], f5 = wq[5];
        const int4v child = reinterpret_cast<const int4v *>(wq)[6];
        const uint32_t o = reinterpret_cast<const uint32_t *>(wq)[28];
        flt4v g0 = f0, g1 = f1, g2 = f2, g3 = f3, g4 = f4, g5 = f5;
        int4v child2 = child;
        uint32_t o2 = o;
        if (PAIR && second != WIDE_NONE) {
          const flt4v *wq2 = reinterpret_cast<const flt4v *>(sc.wide + second);
          g0 = wq2[0]; g1 = wq2[1]; g2 = wq2[2]; g3 = wq2[3]; g4 = wq2[4]; g5 = wq2[5];
          child2 = reinterpret_cast<const int4v *>(wq2)[6];
          o2 = reinterpret_cast<const uint32_t *>(wq2)[28];
        }
        // Visiting order of the binary ordered walk: the top node's two children, nearer first,
        // and within each its own two children, nearer first.  order: 3 bits per binary node
        // (top, first child, second child): axis, and bit 2 = the second child is the lower one.
        auto visit = [&](flt4v a0, flt4v a1, flt4v a2_, flt4v a3, flt4v a4, flt4v a5, int4v ch, uint32_t ord) {
          const float flo[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2_.x, a2_.y, a2_.z, a2_.w};
          const float fhi[12] = {a3.x, a3.y, a3.z, a3.w, a4.x, a4.y, a4.z, a4.w, a5.x, a5.y, a5.z, a5.w};
          auto kid = [&](int i) { return i == 0 ? ch.x : i == 1 ? ch.y : i == 2 ? ch.z : ch.w; };
          n_node += 4;
          uint32_t hit = 0;
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const double lo[3] = {(double)flo[3 * c], (double)flo[3 * c + 1], (double)flo[3 * c + 2]};
            const double hi[3] = {(double)fhi[3 * c], (double)fhi[3 * c + 1], (double)fhi[3 * c + 2]};
            if (box_hit_v(lo, hi, ray, t_min, closest_t)) hit |= 1u << c;
          }
          const uint32_t up = w.up;
          const bool t2 = (((ord >> 2) & 1u) != 0) == (((up >> (ord & 3u)) & 1u) != 0);
          const bool a2 = (((ord >> 5) & 1u) != 0) == (((up >> ((ord >> 3) & 3u)) & 1u) != 0);
          const bool b2 = (((ord >> 8) & 1u) != 0) == (((up >> ((ord >> 6) & 3u)) & 1u) != 0);
          const int an = a2 ? 1 : 0, bn = b2 ? 3 : 2;  // the nearer slot of each pair
          const int s0 = t2 ? bn : an, s1 = s0 ^ 1, s2 = t2 ? an : bn, s3 = s2 ^ 1;
          // farthest first onto the stack, so that the nearest comes off first
          if (hit & (1u << s3)) push(kid(s3));
          if (hit & (1u << s2)) push(kid(s2));
          if (hit & (1u << s1)) push(kid(s1));
          if (hit & (1u << s0)) push(kid(s0));
        };
        if (PAIR && second != WIDE_NONE) visit(g0, g1, g2, g3, g4, g5, child2, o2);
        visit(f0, f1, f2, f3, f4, f5, child, o);
      } else {
        const dbl2 *lq = reinterpret_cast<const dbl2 *>(sc.leaves + ~cur);
        const dbl2 l0 = lq[0], l1 = lq[1], l2 = lq[2];
        const int4v lp = reinterpret_cast<const int4v *>(lq)[3];
        const double lb[6] = {l0.x, l0.y, l1.x, l1.y, l2.x, l2.y};
        auto lprim = [&](int i) { return i == 0 ? lp.x : i == 1 ? lp.y : i == 2 ? lp.z : lp.w; };
        n_node++;
        if (box_hit_v(lb, lb + 3, ray, t_min, closest_t)) {
          if (lp.w != LEAF_INDIRECT && (room >= 4 || (room == 2 ? lp.z : lp.w) < 0)) {  // room >= 2 here
            // up to four primitives named in the record, and room for them
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const int tag = lprim(k);
              if (tag >= 0 && !(ANYHIT && (tag & LEAF_CLEAR_BIT))) {
                const int slot = __ffs(~w.used) - 1;
                pend[slot][tid] = tag & LEAF_PRIM_MASK;
                w.used |= 1u << slot;
                w.km |= (1ull << slot) << (8 * (tag >> LEAF_KIND_SHIFT));
              }
            }
          } else {
            // no room for them all, or a leaf of more than four (trees cut off at the depth limit): test here
            const bool indirect = lp.w == LEAF_INDIRECT;
            const int n = indirect ? lp.y : 4;
            for (int k = 0; k < n; k++) {
              const int tag = indirect ? rdonly(sc.leaf_tagged)[lp.x + k] : lprim(k);
              if (tag >= 0 && !(ANYHIT && (tag & LEAF_CLEAR_BIT))) {
                test(tag & LEAF_PRIM_MASK);
              }
            }
          }
        }
      }
      w.cur = w.sp > 0 ? pop() : WIDE_NONE;
      if (ANYHIT && best.prim >= 0 && best.t < any_limit) w.stop();
    }
    return true;
  }
  // test a batch: every lane that holds a primitive of the kind most lanes hold tests one
  const uint32_t mine = (uint32_t)(w.km >> (8 * kind)) & 0xffu;
  if (mine) {
    const int slot = __ffs(mine) - 1;
    const int pi = pend[slot][tid];
    w.used &= ~(1u << slot);
    w.km &= ~((1ull << slot) << (8 * kind));
    test(pi);
    if (ANYHIT && best.prim >= 0 && best.t < any_limit) w.stop();
  }
  return true;
}

// The rays of the wave whose wide walk met two surfaces within the tie band (`again`; rare: coincident
// surfaces) are walked once more through the binary tree in the reference's own order (traverse_lane), from
// the start: which of the two surfaces wins is then decided as Bvh.cpp:312-332 decides it.  Wave-uniform call.
template <bool UV, bool MOTION>
DEV void walk_again_in_order(const DScene &sc, const RayL &ray, bool again, double t_min, double t_max, Best<UV> &best,
                             unsigned &n_node, unsigned &n_prim) {
  double closest = t_max;
  if (again) {
    best.prim = -1;
    best.t = 0;
  }
  traverse_lane<false, UV, MOTION>(sc, ray, again, t_min, closest, best, n_node, n_prim);
}

template <bool SHADOW, bool UV, bool MOTION>
DEV void traverse_wide(const DScene &sc, const RayL &ray, bool active, double t_min, double &closest_t,
                       Best<UV> &best, unsigned &n_node, unsigned &n_prim) {
  WideWalk w;
  // private memory; touched only by lanes whose stack outgrows LDS (the last two words: the distance limit
  // the walk was called with, kept for walk_again_in_order without holding registers)
  int32_t deep[WIDE_STACK_DEEP + 2];
  if (!SHADOW) {
    const unsigned long long bits = __double_as_longlong(closest_t);
    *(volatile int32_t *)&deep[WIDE_STACK_DEEP] = (int32_t)(bits & 0xffffffffull);
    *(volatile int32_t *)&deep[WIDE_STACK_DEEP + 1] = (int32_t)(bits >> 32);
  }
  w.start(ray, active);
  while (wide_step<SHADOW, UV, MOTION>(sc, ray, t_min, closest_t, best, w, deep, n_node, n_prim)) {
  }
  if (!SHADOW) {
    if (sc.tie_hi > 1.0 && __any(w.tied())) {  // (no band: no tree in the reference's order to walk)
      const unsigned long long bits = (unsigned long long)(uint32_t)*(volatile int32_t *)&deep[WIDE_STACK_DEEP] |
                                      ((unsigned long long)(uint32_t)*(volatile int32_t *)&deep[WIDE_STACK_DEEP + 1] << 32);
      walk_again_in_order<UV, MOTION>(sc, ray, w.tied(), t_min, __longlong_as_double((long long)bits), best, n_node, n_prim);
    }
    if (best.prim >= 0) closest_t = best.t;  // (the walk kept it a tie band beyond)
  }
}

// WALK: WALK_PACKET (small scenes, coherent rays), WALK_LANE (per lane, the reference's order),
// WALK_ORDERED (per lane, nearer child first) or WALK_WIDE (4-wide nodes, nearer first, tests batched by kind).
template <int WALK, bool SHADOW, bool UV, bool MOTION>
DEV void traverse(const DScene &sc, const RayL &ray, bool active, double t_min, double &closest_t, Best<UV> &best,
                  unsigned &n_node, unsigned &n_prim) {
  if (WALK == WALK_PACKET) traverse_packet<SHADOW, UV, MOTION>(sc, ray, active, t_min, closest_t, best, n_node, n_prim);
  else if (WALK == WALK_LANE) traverse_lane<SHADOW, UV, MOTION>(sc, ray, active, t_min, closest_t, best, n_node, n_prim);
  else if (WALK == WALK_ORDERED) traverse_ordered<SHADOW, UV, MOTION>(sc, ray, active, t_min, closest_t, best, n_node, n_prim);
  else traverse_wide<SHADOW, UV, MOTION>(sc, ray, active, t_min, closest_t, best, n_node, n_prim);
}

// What shading needs from a hit (HitRecord, HitRecord.h:8-32).
template <bool UV> struct Surface;
template <> struct Surface<false> {
  d3 p, n;
  bool front;
};
template <> struct Surface<true> {
  d3 p, n;
  bool front;
  double u, v;
  d3 tan, bit;
};

DEV void finish_frame(Surface<false> &, const Best<false> &, bool) {}
DEV void finish_frame(Surface<true> &s, const Best<true> &b, bool plane) {
  s.u = b.u();
  s.v = b.v();
  if (plane) {
    s.tan = b.tan();  // Plane.cpp:124
  } else {
    d3 t = unit(unit(b.tan()));  // Transform.cpp:112-114 then e.g. Sphere.cpp:72
    s.tan = unit(t - s.n * dot(t, s.n));
  }
  s.bit = unit(cross(s.n, s.tan));
}

// Normalisation and face-forwarding of the winning hit (Transform.cpp:107-109, HitRecord.h:28-31).
template <bool UV> DEV Surface<UV> make_surface(const DScene &sc, const RayL &ray, const Best<UV> &b) {
  Surface<UV> s;
  const bool plane = sc.prims[b.prim].type == MIRT_PLANE;
  const d3 bn = b.n();
  const d3 n = plane ? bn : unit(unit(bn));
  s.front = dot(ray.d, n) < 0;
  s.n = s.front ? n : -n;
  s.p = b.p();
  finish_frame(s, b, plane);
  return s;
}

// TextureManager::sample (Texture.cpp:74-150): clamp, flip v, bilinear with wrapped neighbours.
DEV d3 tex_sample(const DTexture &tx, double u, double v) {
  const int w = tx.width, h = tx.height;
  u = maxd(0.0, mind(1.0, u));
  v = maxd(0.0, mind(1.0, v));
  v = 1.0 - v;
  const double x = u * w - 0.5, y = v * h - 0.5;
  const int xf = (int)floor(x), yf = (int)floor(y);
  const double ur = x - xf, vr = y - yf;
  const int x0 = (xf % w + w) % w, x1 = (x0 + 1) % w;
  const int y0 = (yf % h + h) % h, y1 = (y0 + 1) % h;
  const double mv = (double)tx.max_value;
  auto px = [&](int yy, int xx) {
    const uint16_t *q = tx.rgb + 3 * ((size_t)yy * w + xx);
    return mk((double)q[0] / mv, (double)q[1] / mv, (double)q[2] / mv);
  };
  const d3 c00 = px(y0, x0), c10 = px(y0, x1), c01 = px(y1, x0), c11 = px(y1, x1);
  const d3 top = c00 + (c10 - c00) * ur;
  const d3 bot = c01 + (c11 - c01) * ur;
  return top + (bot - top) * vr;
}

// apply_normal_map (Raytracer.cpp:220-242)
DEV d3 shading_normal(const DScene &, const mirt_config &, const mirt_material &, const Surface<false> &s) {
  return s.n;
}
DEV d3 shading_normal(const DScene &sc, const mirt_config &cfg, const mirt_material &m, const Surface<true> &s) {
  if (cfg.enable_textures && m.normal_map >= 0) {
    const d3 ns = tex_sample(sc.textures[m.normal_map], s.u, s.v);
    d3 tn = mk(ns.x * 2.0 - 1.0, ns.y * 2.0 - 1.0, ns.z * 2.0 - 1.0);
    tn.x *= m.bump_strength;
    tn.y *= m.bump_strength;
    tn = unit(tn);
    return unit(s.tan * tn.x + s.bit * tn.y + s.n * tn.z);
  }
  if (cfg.enable_textures && m.bump_map >= 0) {
    const DTexture &tx = sc.textures[m.bump_map];
    const double delta = 0.001;
    auto height = [&](double u, double v) {
      const d3 c = tex_sample(tx, u, v);
      return 0.299 * c.x + 0.587 * c.y + 0.114 * c.z;
    };
    const double hc = height(s.u, s.v);
    const double dU = (height(s.u + delta, s.v) - hc) / delta;
    const double dV = (height(s.u, s.v + delta) - hc) / delta;
    const double scale = 10.0 * m.bump_strength;
    return unit(s.n - s.tan * dU * scale - s.bit * dV * scale);
  }
  return s.n;
}
DEV bool base_texture(const DScene &, const mirt_config &, const mirt_material &, const Surface<false> &, d3 &) {
  return false;
}
DEV bool base_texture(const DScene &sc, const mirt_config &cfg, const mirt_material &m, const Surface<true> &s,
                      d3 &tex) {
  if (cfg.enable_textures && m.has_texture && m.texture >= 0) {  // Raytracer.cpp:250-251, :391-392
    tex = tex_sample(sc.textures[m.texture], s.u, s.v);
    return true;
  }
  return false;
}

// Camera::get_ray (Camera.cpp:10-111); the camera basis comes pre-computed in DCamera.
DEV void camera_ray(const DCamera &c, double x, double y, int width, int height, uint64_t key, d3 &origin, d3 &dir) {
  const double ndc_x = x / width, ndc_y = y / height;
  const double vx = (ndc_x - 0.5) * c.viewport_w, vy = (0.5 - ndc_y) * c.viewport_h;
  const d3 loc = ld3(c.location), w = ld3(c.w), u = ld3(c.u), v = ld3(c.v);
  const d3 image_point = loc - w * c.image_plane_dist + u * vx + v * vy;
  origin = loc;
  dir = unit(image_point - loc);
  if (c.dof) {
    const double t = c.focus_distance / dot(ld3(c.gaze), dir);
    const d3 focus = origin + dir * t;
    const double r1 = mirt_rng_draw(key, MIRT_DIM_LENS_R1), r2 = mirt_rng_draw(key, MIRT_DIM_LENS_R2);
    const double r = c.aperture_radius * sqrt(r1);
    double sn, cs;
    sincos(2.0 * M_PI * r2, &sn, &cs);
    const d3 lens = loc + u * (r * cs) + v * (r * sn);
    origin = lens;
    dir = unit(focus - lens);
  }
}

// Light::sample_point (Light.h:38-69); the basis around the light's normal comes from DLightFrame
// (device_light_basis in engine.cpp evaluates the same expressions once per light).
template <class L> DEV d3 light_point(const L &l, const DLightFrame &lf, double u, double v) {
  if (l.type != MIRT_LIGHT_AREA) return ld3(l.location);
  const d3 right = ld3(lf.right), up = ld3(lf.up);
  if (l.area_shape == MIRT_AREA_RECT)
    return ld3(l.location) + right * ((u - 0.5) * l.area_size[0]) + up * ((v - 0.5) * l.area_size[1]);
  const double r = sqrt(u);
  double sn, cs;
  sincos(2.0 * 3.14159265359 * v, &sn, &cs);
  return ld3(l.location) + right * (r * cs * (l.area_size[0] * 0.5)) + up * (r * sn * (l.area_size[1] * 0.5));
}

// What a shadow item needs of its light before it can start, for the first LIGHT_TABLE_MAX lights of
// the scene, in LDS: read from memory, these are two more dependent round trips after the item's own
// (the item names the light, the light's records hold the rest), and the set-up of an item is a third
// of the shadow kernel's time.  Lights beyond the table are read from memory as before.
#define LIGHT_TABLE_MAX 32
struct LightTable {
  double loc[LIGHT_TABLE_MAX][3];
  double inv_n[LIGHT_TABLE_MAX];
  uint32_t dim[LIGHT_TABLE_MAX];
  int32_t side[LIGHT_TABLE_MAX];
  int32_t type[LIGHT_TABLE_MAX];
};
DEV void light_table_fill(LightTable &t, const DScene &sc, const DFrame &fr) {
  for (int i = (int)threadIdx.x; i < sc.n_lights && i < LIGHT_TABLE_MAX; i += (int)blockDim.x) {
    const DLightFrame lf = fr.light_frame[i];
    t.loc[i][0] = sc.lights[i].location[0];
    t.loc[i][1] = sc.lights[i].location[1];
    t.loc[i][2] = sc.lights[i].location[2];
    t.inv_n[i] = lf.inv_n;
    t.dim[i] = lf.dim;
    t.side[i] = lf.side;
    t.type[i] = sc.lights[i].type;
  }
  __syncthreads();
}
DEV int light_side(const LightTable &t, const DFrame &fr, uint32_t li) {
  return li < LIGHT_TABLE_MAX ? t.side[li] : fr.light_frame[li].side;
}
DEV double light_inv_n(const LightTable &t, const DFrame &fr, uint32_t li) {
  return li < LIGHT_TABLE_MAX ? t.inv_n[li] : fr.light_frame[li].inv_n;
}

// The point of stratum s (of side x side) on light li, for the sample whose key is `key`
// (Raytracer.cpp:160-170).  A light that is not an area light has one point and spends no draw on it.
DEV d3 stratum_point(const LightTable &t, const DScene &sc, const DFrame &fr, uint32_t li, uint64_t key, int s, int side) {
  const bool tabled = li < LIGHT_TABLE_MAX;
  const int type = tabled ? t.type[li] : sc.lights[li].type;
  if (type != MIRT_LIGHT_AREA) return tabled ? mk(t.loc[li][0], t.loc[li][1], t.loc[li][2]) : ld3(sc.lights[li].location);
  const int si = s / side, sj = s % side;
  const uint32_t dim = (tabled ? t.dim[li] : fr.light_frame[li].dim) + 2u * (uint32_t)s;
  const double u = (si + mirt_rng_draw(key, dim)) / side;
  const double v = (sj + mirt_rng_draw(key, dim + 1)) / side;
  return light_point(sc.lights[li], fr.light_frame[li], u, v);
}

// The accumulator is kept as three planes, not as pixel triples: the 64 pixels a wave adds to are mostly
// neighbours, and one atomic instruction over neighbours in a plane touches a third of the cache lines
// it would touch across triples (the atomic units work a line at a time).  Adding zero is skipped.
DEV void pixel_add(double *accum, uint32_t plane, uint32_t pixel, d3 c) {
  double *p = accum + pixel;
  if (c.x != 0.0) unsafeAtomicAdd(p, c.x);
  if (c.y != 0.0) unsafeAtomicAdd(p + plane, c.y);
  if (c.z != 0.0) unsafeAtomicAdd(p + 2 * (size_t)plane, c.z);
}

// The same for all lanes of a wave at once (`on`: this lane has something to add), with the terms of
// neighbouring lanes that go to the same pixel summed in the wave first.  Rays that a glossy hit spawns sit
// next to one another in their queue (eight siblings at the first bounce, sixty-four cousins at the second)
// and so do their shadow items: without this a pixel behind two glossy bounces takes a thousand adds to
// one address per frame, and adds to one address go one at a time -- measured 40 to 70 microseconds per
// launch of the deep levels, whatever the launch carries.  Costs a ballot when no two neighbours share a pixel.
// (Measured on Test5 with 8 glossy samples, depth 8: an eighth of the 1080p frame 2.15 -> 1.69 ms, the whole
// frame 10.4 -> 10.9 ms; summing only runs longer than eight lanes: 1.94 and 11.8 ms.)
// Must be called by all lanes of the wave.
DEV void pixel_add_wave(double *accum, uint32_t plane, uint32_t pixel, d3 c, bool on) {
  const int lane = __lane_id();
  const uint32_t key = on ? pixel : 0xffffffffu;
  const uint32_t prev = __shfl_up(key, 1);
  const bool head = lane == 0 || key != prev || !on;
  const unsigned long long heads = __ballot(head);
  if (heads == ~0ull) {  // no run longer than one lane
    if (on) pixel_add(accum, plane, pixel, c);
    return;
  }
  if (!on) c = mk(0, 0, 0);
  // bit i of `more`: lanes i + 1 .. i + off all continue lane i's run; doubling steps while any run is that long
  unsigned long long more = ~heads >> 1;
  for (int off = 1; more != 0; off <<= 1) {
    const double x = __shfl_down(c.x, off), y = __shfl_down(c.y, off), z = __shfl_down(c.z, off);
    if ((more >> lane) & 1ull) {
      c.x += x;
      c.y += y;
      c.z += z;
    }
    more &= more >> off;
  }
  if (on && head) pixel_add(accum, plane, pixel, c);
}

// Appending to the block's shard of a ray queue: the wave counts what it is going to append, one lane
// reserves the range with a single atomic on the shard's tail (queue_reserve), and every pushing
// lane writes at the range's start + its rank among the pushers (queue_write).
DEV uint32_t queue_reserve(uint32_t *count, uint32_t shard, uint32_t n) {
  uint32_t base = 0;
  if (__lane_id() == 0) base = atomicAdd(count + shard * QUEUE_COUNT_STRIDE, n);
  return __shfl(base, 0);
}
DEV void queue_write(const DRayQueue &q, uint32_t shard, uint32_t r, d3 o, d3 d, double time, d3 w, uint64_t key,
                     uint32_t pixel) {
  if (r < q.shard_cap) {  // a full shard drops the ray; the host sees the tail past the capacity and redoes the frame
    const size_t i = (size_t)shard * q.shard_cap + r;
    q.ox[i] = o.x; q.oy[i] = o.y; q.oz[i] = o.z;
    q.dx[i] = d.x; q.dy[i] = d.y; q.dz[i] = d.z;
    q.time[i] = time;
    q.wr[i] = w.x; q.wg[i] = w.y; q.wb[i] = w.z;
    q.key[i] = key;
    q.pixel[i] = pixel;
  }
}

DEV void shadow_store(DShadowItem *items, size_t i, d3 p, d3 c, uint64_t key, uint32_t pixel, uint32_t light) {
  DShadowItem it;
  it.p[0] = p.x; it.p[1] = p.y; it.p[2] = p.z;
  it.c[0] = c.x; it.c[1] = c.y; it.c[2] = c.z;
  it.key = key;
  it.pixel = pixel;
  it.light = light;
  items[i] = it;
}

// A consumer's view of a sharded queue, held across the lanes of a wave: lane s < QUEUE_SHARDS knows
// shard s's entry count and the first virtual block (256 entries, never straddling two shards) that
// works on it.  Every wave of the grid derives the same view from the producer's tail counters.
struct WaveView {
  uint32_t begin, blocks, len;
  uint32_t total;  // virtual blocks over all shards (wave-uniform)
};
DEV WaveView wave_view(const uint32_t *count, uint32_t shard_cap) {
  const int lane = __lane_id();
  WaveView v;
  v.len = 0;
  if (lane < QUEUE_SHARDS) {
    const uint32_t c = count[lane * QUEUE_COUNT_STRIDE];
    v.len = c < shard_cap ? c : shard_cap;
  }
  v.blocks = (v.len + 255u) >> 8;
  uint32_t incl = v.blocks;
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  v.begin = incl - v.blocks;
  v.total = __builtin_amdgcn_readlane(incl, WAVE - 1);
  return v;
}
// The entry lane `threadIdx.x` of virtual block `vb` works on; false for lanes past the shard's end.
DEV bool view_entry(const WaveView &v, uint32_t vb, uint32_t shard_cap, size_t &slot) {
  const unsigned long long m = __ballot(vb >= v.begin && vb < v.begin + v.blocks);
  const int s = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
  const uint32_t begin = __builtin_amdgcn_readlane(v.begin, s), len = __builtin_amdgcn_readlane(v.len, s);
  const uint32_t i = (vb - begin) * 256u + threadIdx.x;
  slot = (size_t)s * shard_cap + (i < len ? i : 0);
  return i < len;
}

// The replica of the statistics counters this wave adds to.
DEV DCounters *my_counters(DCounters *c) { return c + ((blockIdx.x * 4u + (threadIdx.x >> 6)) % COUNTER_SHARDS); }

DEV void count_add(unsigned long long *dst, unsigned v) {
  // wave sum, then one atomic per wave
  unsigned s = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (__lane_id() == 0 && s) atomicAdd(dst, (unsigned long long)s);
}

DEV d3 reflect_dir(d3 i, d3 n) { return i - n * 2.0 * dot(i, n); }  // Raytracer.cpp:210-212


// Pixel of lane `p` (0..255) of 16x16 block `block` of this rank's tiles: four 8x8 waves per block.
DEV bool block_pixel(const DFrame &fr, uint32_t block, uint32_t p, int &gx, int &gy, uint32_t &slot_pixel) {
  const uint32_t per_tile = (uint32_t)(fr.blocks_x * fr.blocks_y);
  const uint32_t slot = block / per_tile, b = block % per_tile;
  const int tile = tile_of_slot((int)slot, fr.rank, fr.tiles_x, fr.world, fr.stagger);
  const int tx = tile % fr.tiles_x, ty = tile / fr.tiles_x;
  const uint32_t wv = p >> 6, ln = p & 63;
  const int lx = (int)((b % fr.blocks_x) * 16 + (wv & 1) * 8 + (ln & 7));
  const int ly = (int)((b / fr.blocks_x) * 16 + (wv >> 1) * 8 + (ln >> 3));
  gx = tx * fr.tile_w + lx;
  gy = ty * fr.tile_h + ly;
  slot_pixel = slot * (uint32_t)(fr.tile_w * fr.tile_h) + (uint32_t)(ly * fr.tile_w + lx);
  return lx < fr.tile_w && ly < fr.tile_h && gx < fr.width && gy < fr.height;
}

// One block of 256 rays: virtual block `vb` of the launch.  `active` and `j` locate the lane's queue
// entry (unused by PRIMARY, whose virtual blocks are (16x16 pixel block, sample index) pairs).
// Progress feed (the reference times every image row, Raytracer.cpp:527,573-584): when asked for, a
// primary block notes the wall clock when it starts and when its samples are shaded, per band of 16
// pixel rows.  One lane per block, two atomics; nothing when a.progress is NULL.
// The 16x16 block that virtual block `vb` of a primary launch renders (see TraceArgs::first); false: none.
DEV bool primary_block(const TraceArgs &a, uint32_t vb, uint32_t &block) {
  const uint32_t k = vb / (uint32_t)a.fr.spp, run = k / a.group;
  block = (a.first + run * a.stride) * a.group + (k - run * a.group);
  if (block < a.total) return true;
  block = 0;
  return false;
}
DEV void progress_mark(const TraceArgs &a, uint32_t vb, bool finished) {
  if (!a.progress || threadIdx.x != 0) return;
  uint32_t block;
  if (!primary_block(a, vb, block)) return;
  int gx, gy;
  uint32_t px;
  block_pixel(a.fr, block, 0, gx, gy, px);
  if (gy >= a.fr.height) return;  // a block wholly below the image (an edge tile's padding) has no row band
  const unsigned long long now = wall_clock64();
  if (finished) atomicMax(a.progress + 1 + gy / 16, now);
  else atomicMin(a.progress, now);
}

// The kernel's argument block, read afresh.  The compiler loads every argument field a kernel uses at its
// entry and, out of scalar registers long before the walk is over, parks them in lanes of vector registers and
// fetches them back one by one when the shading needs them (a hundred lane writes at the top of k_trace, several
// hundred lane reads in its shading section).  A pointer to the argument segment that has passed through an
// empty asm is a pointer the compiler knows nothing about: what is read through it is read after that point,
// from the scalar cache, when it is needed.  Only for kernels whose one parameter is the block itself.
#ifndef MIRT_ARGS_AGAIN
#define MIRT_ARGS_AGAIN 1
#endif
template <class A> DEV const A &args_again(const A &a) {
#if MIRT_ARGS_AGAIN
  const RO A *p = (const RO A *)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const A *)p;
#else
  return a;
#endif
}

// One ray of a trace launch: where it comes from.
struct RaySeed {
  d3 ro, rd, W;  // origin, direction, the RGB weight with which its radiance enters the pixel
  double time;
  uint64_t key;
  uint32_t pixel;
};
// PRIMARY: the camera ray of virtual block `vb` = (16x16 pixel block, sample index), lane `p` of the
// block (Raytracer.cpp:533-541); returns false for lanes outside the image.
DEV bool primary_ray(const TraceArgs &a, uint32_t vb, uint32_t p, RaySeed &r) {
  const mirt_config &cfg = a.fr.cfg;
  uint32_t block;
  const bool exists = primary_block(a, vb, block);
  const uint32_t s = vb % (uint32_t)a.fr.spp;
  int gx, gy;
  const bool active = block_pixel(a.fr, block, p, gx, gy, r.pixel) && exists;
  r.key = mirt_rng_sample_key(cfg.seed, (uint32_t)(gy * a.fr.width + gx), s);
  double u, v;
  if (cfg.sampling == MIRT_SAMPLING_CENTRE) {  // deterministic parity mode
    u = gx + 0.5;
    v = gy + 0.5;
    r.time = 0.5;
  } else {  // Raytracer.cpp:533-538
    u = gx + mirt_rng_draw(r.key, MIRT_DIM_JITTER_U);
    v = gy + mirt_rng_draw(r.key, MIRT_DIM_JITTER_V);
    r.time = cfg.enable_motion_blur ? mirt_rng_draw(r.key, MIRT_DIM_TIME) : 0.5;
  }
  camera_ray(a.fr.cam, u, v, a.fr.width, a.fr.height, r.key, r.ro, r.rd);
  r.W = mk(1, 1, 1);
  return active;
}
// entry j of a ray queue
DEV void queued_ray(const DRayQueue &in, size_t j, RaySeed &r) {
  r.ro = mk(in.ox[j], in.oy[j], in.oz[j]);
  r.rd = mk(in.dx[j], in.dy[j], in.dz[j]);
  r.time = in.time[j];
  r.W = mk(in.wr[j], in.wg[j], in.wb[j]);
  r.key = in.key[j];
  r.pixel = in.pixel[j];
}

// Unshadowed Blinn-Phong contribution of one light to the pixel (Raytracer.cpp:266-285): `to` from the
// point to the light, sn the shading normal, rd the ray direction; wd / ws the weights with which the
// diffuse and the specular term enter the pixel.  Colour terms only: one reciprocal per normalisation
// instead of three divisions.
template <class L>
DEV d3 light_term(const L &light, const mirt_config &cfg, d3 to, d3 sn, d3 rd, d3 wd, d3 ws, d3 tint, d3 specular,
                  double shininess) {
  const double inv_d = rsqrt_shade(len2(to));
  const d3 ldir = to * inv_d;
  const double ndl = maxd(0.0, dot(sn, ldir));
  const d3 lin = ld3(light.color) * (light.intensity * cfg.light_intensity_factor * (inv_d * inv_d));
  d3 c = wd * (tint * ndl) * lin;
  const d3 hv = ldir + (-rd);
  const double h2 = len2(hv);  // zero only when the light is exactly behind the eye ray
  const double ndh = h2 > 0.0 ? maxd(0.0, dot(sn, hv) * rsqrt_shade(h2)) : 0.0;
  if (ndh > 0.0) c = c + ws * (specular * pow_shade(ndh, shininess)) * lin;
  return c;
}

// What differs from level to level of a frame's ray tree (wave-uniform).
struct LevelVar {
  uint32_t *out_count;     // tails of the queue this level's hits append rays to (unused when they spawn nothing)
  uint32_t *shadow_count;  // tails of this level's shadow items
  uint32_t shadow_base;    // where this level's items start in the block's shard of the shadow queue
  int32_t depth;
};
DEV LevelVar level_of(const TraceArgs &a) { return LevelVar{a.out_count, a.shadow_count, 0u, a.depth}; }

// What trace() does with the closest hit (Raytracer.cpp:305-472), for the lanes of the wave with
// `active` set: background for a miss, local terms without visibility, one shadow work item per
// contributing light, and the rays the hit spawns.  Must be called by all lanes of the wave; the
// appends go to shard `shard` of the launch's queues.
template <bool UV>
DEV void shade_hits(const TraceArgs &a, const DRayQueue &out, const LevelVar &lv, uint32_t shard, const RayL &ray,
                    const Best<UV> &best, bool active, d3 W, uint64_t key, uint32_t pixel) {
  const DScene &sc = a.sc;
  const mirt_config &cfg = a.fr.cfg;
  const double time = ray.time;
  const bool hit = active && best.prim >= 0;
  d3 add = mk(0, 0, 0);  // what this ray adds to its pixel here: the background, or the hit's local terms
  bool adds = false;
  if (active && !hit) {  // Raytracer.cpp:472
    add = W * ld3(sc.background);
    adds = true;
  }

  // ---- shading of the hit (Raytracer.cpp:305-466), visibility deferred to k_shadow ------------
  const bool spawn = lv.depth + 1 < cfg.max_ray_depth;  // deeper rays would return black (Raytracer.cpp:297)
  // lanes without a hit bind `m` to material 0 and never read it (a scene may have no primitives at all)
  const mirt_material &m = sc.materials[hit ? sc.prims[best.prim].material : 0];
  Surface<UV> sf = {};
  d3 sn = mk(0, 0, 1), tint = mk(0, 0, 0), amb = mk(0, 0, 0);
  double R = 0, T = 0, eps = 0;
  bool pass = false, glass = false, tir = false;
  double fres = 0;
  d3 perp = mk(0, 0, 0);
  double disc = 0;
  d3 wd = mk(0, 0, 0), ws = mk(0, 0, 0);
  bool lit = false;
  if (hit) {
    sf = make_surface<UV>(sc, ray, best);
    sn = shading_normal(sc, cfg, m, sf);
    R = m.reflectivity;
    T = m.transparency;
    tint = ld3(m.diffuse);
    amb = ld3(m.ambient);
    d3 tex;
    if (base_texture(sc, cfg, m, sf, tex)) {
      tint = tex * ld3(m.diffuse);
      amb = tex * ld3(m.ambient);
    }
    eps = cfg.ray_offset_epsilon;
    if (cfg.use_adaptive_epsilon) eps += len(sf.p) * cfg.adaptive_epsilon_scale;
    glass = T > 0.0;
    if (glass) {  // Raytracer.cpp:414-464
      const double eta = sf.front ? (1.0 / m.refractive_index) : m.refractive_index;
      const double ct = fabs(dot(ray.d, sf.n));
      double r0 = (eta - 1.0) / (eta + 1.0);
      r0 = r0 * r0;
      const double g = 1.0 - ct, g2 = g * g;
      fres = r0 + (1.0 - r0) * (g2 * g2 * g);  // Schlick's fifth power: a colour weight, not geometry
      perp = (ray.d + sf.n * ct) * eta;
      disc = 1.0 - len2(perp);
      tir = !(disc >= 0);
      pass = !tir && T >= 0.99;  // `return combined`: every local term is dropped (Raytracer.cpp:456-458)
    }
    if (!pass) {
      const double kT = glass ? (1.0 - T) : 1.0;
      ws = W * kT;
      wd = (R > 0.0) ? ws * (1.0 - R) : ws;  // Raytracer.cpp:408-410
      d3 local = ld3(m.emission) * m.emission_strength;
      local = W * local;
      lit = !(T >= cfg.pure_glass_threshold);  // pure glass is not shaded (Raytracer.cpp:307-311)
      if (lit) local = local + wd * (amb * cfg.ambient_factor);
      add = local;
      adds = true;
    }
  }
  pixel_add_wave(a.accum, a.fr.plane, pixel, add, adds);

  // per-light unshadowed Blinn-Phong terms -> shadow work items (Raytracer.cpp:262-286).  A light
  // that faces away from the shading normal adds exactly zero in the reference, so it gets no item.
  // First pass counts the wave's items so that ONE atomic reserves room for all of them; it decides
  // "faces the light" on the un-normalised direction (same sign as the reference's N.L up to the last
  // rounding of a product that is then of the order of 1e-16 anyway) and the second pass keeps to it.
  //
  // The rays this wave is going to append (mirror, every glossy-lobe sample, Fresnel reflection,
  // refraction) are counted here too, and the two reservations -- shadow items, rays -- are issued
  // together: each is an atomic whose result the wave has to wait for, and one wait covers both.
  const bool refl = spawn && hit && R > 0.0 && !pass;
  // a lane without a hit has no material: `m` is only read behind `refl`
  int samples = 1;
  if (refl && cfg.glossy_samples > 1 && m.glossiness < 0.94 && lv.depth < 2) samples = cfg.glossy_samples;
  const bool mirror = samples == 1 || (refl && m.glossiness > 0.94);
  const bool push_mirror = refl && mirror, lobe = refl && !mirror;  // a lobe lane appends glossy_samples rays
  const bool push_reflect = spawn && hit && glass, push_refract = spawn && hit && glass && !tir;
  const unsigned long long m_mirror = __ballot(push_mirror), m_lobe = __ballot(lobe), m_reflect = __ballot(push_reflect),
                           m_refract = __ballot(push_refract);
  const uint32_t n_mirror = (uint32_t)__popcll(m_mirror), n_lobe = (uint32_t)__popcll(m_lobe),
                 n_reflect = (uint32_t)__popcll(m_reflect), n_refract = (uint32_t)__popcll(m_refract);
  const uint32_t n_samples = (uint32_t)cfg.glossy_samples;
  const uint32_t total = n_mirror + n_lobe * n_samples + n_reflect + n_refract;
  uint32_t ray_base = 0;
  if (a.shadow_groups > 1) {
    // Items sorted by light as they are queued (large scenes): light li appends to its own group of
    // a.shadow_per shards, so that the shadow launch -- which works through the queue shard after shard --
    // hands a wave 64 items that look towards the same light from neighbouring hits: their walks touch the
    // same nodes.  One count per light in a byte of `packed` (at most 8 lights, at most 64 items each);
    // lane li reserves light li's room, all lights in one atomic instruction.
    unsigned long long packed = 0;
    for (int li = 0; li < sc.n_lights; li++) {
      const bool push = hit && lit && dot(sn, ld3(rdonly(sc.lights)[li].location) - sf.p) > 0.0;
      packed |= (unsigned long long)__popcll(__ballot(push)) << (8 * li);
    }
    const int lane = __lane_id();
    const uint32_t sub = shard % a.shadow_per;
    uint32_t my_base = 0;
    if (lane < sc.n_lights) {
      const uint32_t cnt = (uint32_t)(packed >> (8 * lane)) & 0xffu;
      if (cnt) my_base = atomicAdd(lv.shadow_count + ((uint32_t)lane * a.shadow_per + sub) * QUEUE_COUNT_STRIDE, cnt);
    }
    if (lane == 0 && total) ray_base = atomicAdd(lv.out_count + shard * QUEUE_COUNT_STRIDE, total);
    ray_base = __shfl(ray_base, 0);
    if (packed) {
      const unsigned long long below = (1ull << lane) - 1ull;
      for (int li = 0; li < sc.n_lights; li++) {
        const uint32_t base = __shfl(my_base, li);
        if (!((packed >> (8 * li)) & 0xffull)) continue;  // wave-uniform
        const RO mirt_light &L = rdonly(sc.lights)[li];
        const d3 to = ld3(L.location) - sf.p;
        const bool push = hit && lit && dot(sn, to) > 0.0;
        d3 c = mk(0, 0, 0);
        if (push) c = light_term(L, cfg, to, sn, ray.d, wd, ws, tint, ld3(m.specular), m.shininess);
        const uint32_t r = base + (uint32_t)__popcll(__ballot(push) & below);
        if (push && r < a.shadow_shard_cap)  // a full shard drops the item; the host redoes the frame
          shadow_store(a.shadow, (size_t)((uint32_t)li * a.shadow_per + sub) * a.shadow_shard_cap + r, sf.p, c, key, pixel,
                       (uint32_t)li);
      }
    }
  } else {
    uint32_t items = 0;
    for (int li = 0; li < sc.n_lights; li++) {
      const bool push = hit && lit && dot(sn, ld3(rdonly(sc.lights)[li].location) - sf.p) > 0.0;
      items += (uint32_t)__popcll(__ballot(push));
    }
    uint32_t base = 0;
    if (__lane_id() == 0) {
      if (items) base = atomicAdd(lv.shadow_count + shard * QUEUE_COUNT_STRIDE, items);
      if (total) ray_base = atomicAdd(lv.out_count + shard * QUEUE_COUNT_STRIDE, total);
    }
    base = __shfl(base, 0) + lv.shadow_base;
    ray_base = __shfl(ray_base, 0);
    if (items) {
      const unsigned long long below = (1ull << __lane_id()) - 1ull;
      for (int li = 0; li < sc.n_lights; li++) {
        const RO mirt_light &L = rdonly(sc.lights)[li];
        d3 c = mk(0, 0, 0);
        const d3 to = ld3(L.location) - sf.p;
        const bool push = hit && lit && dot(sn, to) > 0.0;
        if (push) c = light_term(L, cfg, to, sn, ray.d, wd, ws, tint, ld3(m.specular), m.shininess);
        const unsigned long long mask = __ballot(push);
        const uint32_t r = base + (uint32_t)__popcll(mask & below);
        if (push && r < a.shadow_shard_cap)  // a full shard drops the item; the host redoes the frame
          shadow_store(a.shadow, (size_t)shard * a.shadow_shard_cap + r, sf.p, c, key, pixel, (uint32_t)li);
        base += (uint32_t)__popcll(mask);
      }
    }
  }

  if (!spawn) return;  // wave-uniform

  // ---- children: reflection (Raytracer.cpp:317-411; dropped when the glass branch returns `combined`),
  // Fresnel reflection and refraction (Raytracer.cpp:414-464), written kind by kind into the room
  // reserved above.
  if (total) {
    const unsigned long long below = (1ull << __lane_id()) - 1ull;
    uint32_t at = ray_base;
    const d3 r = reflect_dir(ray.d, sn);
    const bool metal = R > 0.5 && T < 0.1;
    d3 cw = ws * R;
    if (metal) cw = cw * tint;
    const d3 org = sf.p + sf.n * eps;
    if (push_mirror)
      queue_write(out, shard, at + (uint32_t)__popcll(m_mirror & below), org, r, time, cw,
                  mirt_rng_child_key(key, MIRT_SLOT_GLOSSY), pixel);
    at += n_mirror;
    if (m_lobe) {
      const d3 w = r;
      const d3 u = unit(cross(fabs(w.x) > 0.1 ? mk(0, 1, 0) : mk(1, 0, 0), w));
      const d3 v = cross(w, u);
      // (library pow and sincos on purpose: a lobe direction that differs from the reference's in its last bit
      // sends the ray to the other of two coincident surfaces -- Test4's cylinder on its plane -- in a dozen
      // pixels of a small frame; the short forms were 11 % faster on Test5 with 8 lobe samples)
      const double expo = pow(10.0, (lobe ? m.glossiness : 0.0) * 4.0);
      const d3 cws = cw / (double)samples;
      const uint32_t rank = (uint32_t)__popcll(m_lobe & below);
      for (uint32_t i = 0; i < n_samples; i++) {
        const double r1 = mirt_rng_draw(key, MIRT_DIM_GLOSSY + 2u * i);
        const double r2 = mirt_rng_draw(key, MIRT_DIM_GLOSSY + 2u * i + 1);
        const double ct = pow(r1, 1.0 / (expo + 1.0));
        const double st = sqrt(maxd(0.0, 1.0 - ct * ct));
        double sp, cp;
        sincos(2.0 * M_PI * r2, &sp, &cp);
        d3 dir = unit(u * (st * cp) + v * (st * sp) + w * ct);
        if (dot(dir, sf.n) < 0) dir = r;
        // (a hit's samples next to one another: what they add to the pixel is then summed in the wave first)
        if (lobe) queue_write(out, shard, at + rank * n_samples + i, org, dir, time, cws, mirt_rng_child_key(key, MIRT_SLOT_GLOSSY + i), pixel);
      }
      at += n_lobe * n_samples;
    }
    const d3 wbase = pass ? W : W * T;
    if (push_reflect)
      queue_write(out, shard, at + (uint32_t)__popcll(m_reflect & below), sf.p + sf.n * eps, reflect_dir(ray.d, sf.n), time,
                  tir ? wbase : wbase * fres, mirt_rng_child_key(key, MIRT_SLOT_GLASS_REFLECT), pixel);
    at += n_reflect;
    if (push_refract)
      queue_write(out, shard, at + (uint32_t)__popcll(m_refract & below), sf.p - sf.n * eps,
                  perp + sf.n * (-sqrt(maxd(0.0, disc))), time, wbase * (1.0 - fres),
                  mirt_rng_child_key(key, MIRT_SLOT_GLASS_REFRACT), pixel);
  }
}

// One block of 256 rays: virtual block `vb` of the launch.  `active` and `j` locate the lane's queue
// entry (unused by PRIMARY, whose virtual blocks are (16x16 pixel block, sample index) pairs).
// `in` / `out`: the queue the rays come from (unused by PRIMARY) and the one their hits append to.
template <bool PRIMARY, bool UV, bool MOTION, int WALK>
DEV void trace_block(const TraceArgs &a, const DRayQueue &in, const DRayQueue &out, const LevelVar &lv, uint32_t vb,
                     bool active, size_t j, unsigned &n_node, unsigned &n_prim) {
  const DScene &sc = a.sc;
  const uint32_t shard = vb % QUEUE_SHARDS;  // of the two queues this block appends to
#if MIRT_PROFILE == 1
  PROF_DECL;
#endif
  RaySeed r;
  if (PRIMARY) {
    progress_mark(a, vb, false);
    active = primary_ray(a, vb, threadIdx.x, r);
  } else {
    queued_ray(in, j, r);
  }
  const RayL ray = make_ray(r.ro, r.rd, r.time);
  Best<UV> best;
  BEST_IN_LDS(UV, best);
  double closest = 1.7976931348623157e308;  // numeric_limits<double>::max() (Raytracer.cpp:300)
#if MIRT_PROFILE == 1
  PROF_MARK(0);
#endif
  // the ray's weight, key and pixel wait in LDS while the traversal needs the registers
  __shared__ double park_w[3][256];
  __shared__ uint64_t park_key[256];
  __shared__ uint32_t park_pixel[256];
  park_w[0][threadIdx.x] = r.W.x;
  park_w[1][threadIdx.x] = r.W.y;
  park_w[2][threadIdx.x] = r.W.z;
  park_key[threadIdx.x] = r.key;
  park_pixel[threadIdx.x] = r.pixel;
  PARK_FENCE();
  traverse<WALK, false, UV, MOTION>(args_again(a).sc, ray, active, 1e-5, closest, best, n_node, n_prim);
  PARK_FENCE();
  const d3 W = mk(park_w[0][threadIdx.x], park_w[1][threadIdx.x], park_w[2][threadIdx.x]);
  const uint64_t key = park_key[threadIdx.x];
  const uint32_t pixel = park_pixel[threadIdx.x];
#if MIRT_PROFILE == 1
  PROF_MARK(1);
#endif
  {
    const TraceArgs &as = args_again(a);
    shade_hits<UV>(as, as.out, level_of(as), shard, ray, best, active, W, key, pixel);
  }
  if (PRIMARY) progress_mark(a, vb, true);
#if MIRT_PROFILE == 1
  PROF_MARK(2);
  PROF_FLUSH(my_counters(a.counters));
#endif
}

// PRIMARY: one grid block per (pixel block, sample).  Otherwise the grid has a fixed size and its
// blocks stride over the virtual blocks of the input queue, whose length only the device knows.
template <bool PRIMARY, bool UV, bool MOTION, int WALK>
__global__ __launch_bounds__(256) WAVES_FOR_WALK(WALK, UV ? 3 : MIRT_TRACE_WAVES) void k_trace(const TraceArgs a) {
  unsigned n_node = 0, n_prim = 0;
  const LevelVar lv = level_of(a);
  if (PRIMARY) {
    trace_block<true, UV, MOTION, WALK>(a, a.in, a.out, lv, blockIdx.x, true, 0, n_node, n_prim);
  } else {
    const WaveView v = wave_view(a.in_count, a.in.shard_cap);
    for (uint32_t vb = blockIdx.x; vb < v.total; vb += gridDim.x) {
      size_t j;
      const bool active = view_entry(v, vb, a.in.shard_cap, j);
      trace_block<false, UV, MOTION, WALK>(a, a.in, a.out, lv, vb, active, j, n_node, n_prim);
    }
  }
#ifndef MIRT_PROFILE
  count_add(&my_counters(a.counters)->node_tests, n_node);
  count_add(&my_counters(a.counters)->prim_tests, n_prim);
#endif
}


// A resident wave's view of a sharded queue as one run of entries 0 .. total-1 (shard after shard),
// and its hand-out of entry numbers: chunks of WORK_CHUNK taken off a launch-wide counter with one
// atomic each, dealt to the lanes that ask.
#ifndef REFILL_MIN
#define REFILL_MIN 8  // hand out new work once this many lanes of a wave are idle
#endif
#define WORK_CHUNK 256u
struct QueueRun {
  uint32_t begin, slen;  // lane s < QUEUE_SHARDS: number of shard s's first entry, entries in shard s
  uint32_t total;
  DEV void open(const uint32_t *count, uint32_t shard_cap) {
    const int lane = __lane_id();
    slen = 0;
    if (lane < QUEUE_SHARDS) {
      const uint32_t c = count[lane * QUEUE_COUNT_STRIDE];
      slen = c < shard_cap ? c : shard_cap;
    }
    uint32_t incl = slen;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    begin = incl - slen;
    total = __builtin_amdgcn_readlane(incl, WAVE - 1);
  }
  // slot of entry g in a queue whose shards hold shard_cap entries each
  DEV size_t slot(uint32_t g, uint32_t shard_cap) const {
    uint32_t shard = 0, at = 0;
#pragma unroll 4
    for (int sh = 0; sh < QUEUE_SHARDS; sh++) {
      const uint32_t b0 = __builtin_amdgcn_readlane(begin, sh), l0 = __builtin_amdgcn_readlane(slen, sh);
      if (g >= b0 && g - b0 < l0) {
        shard = (uint32_t)sh;
        at = g - b0;
      }
    }
    return (size_t)shard * shard_cap + at;
  }
};
struct WorkDealer {
  uint32_t next = 0, end = 0;  // the wave's chunk (wave-uniform)
  bool exhausted = false;      // the counter has passed `total` (wave-uniform)
  // Deals entry numbers to the lanes with `want` set; returns true (and the number in g) for the
  // lanes that got one.  Must be called by all lanes of the wave.
  DEV bool deal(uint32_t *grab, uint32_t total, bool want, uint32_t &g) {
    const int lane = __lane_id();
    if (next == end) {
      uint32_t b = 0;
      if (lane == 0) b = atomicAdd(grab, WORK_CHUNK);
      b = __builtin_amdgcn_readfirstlane(b);
      next = b < total ? b : total;
      end = b + WORK_CHUNK < total ? b + WORK_CHUNK : total;
      exhausted = next == end;
    }
    const unsigned long long wm = __ballot(want);
    g = next + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull));
    const bool take = want && g < end;
    const uint32_t n_want = (uint32_t)__popcll(wm);
    next = next + n_want < end ? next + n_want : end;
    return take;
  }
};

// k_trace for the wide walk, split in two so that lanes can take a new ray as soon as theirs has
// found its closest hit (on a large scene a wave is otherwise as slow as its slowest ray, and the
// shading code, if it has to run whenever a few lanes are ready, runs mostly empty):
//   k_closest_stream   resident grid; a lane takes a ray number off the launch's work counter,
//                      generates the camera ray or reads the queued one, walks, writes the hit
//                      (primitive, distance, point, normal[, uv, tangent]) to hits[ray number];
//   k_shade            one lane per ray number, full waves: rebuilds the ray, reads its hit, and does
//                      what trace() does with it (shade_hits) -- no traversal code in this kernel.
// Same rays, same hits, same shading and appends as k_trace.
template <bool UV> DEV void hit_store(DHit *hits, DHitUV *hits_uv, size_t g, const Best<UV> &best) {
  DHit h;
  h.prim = best.prim;
  h.pad = 0;
  h.t = best.t;
  d3 p = mk(0, 0, 0), n = mk(0, 0, 0);
  if (best.prim >= 0) {
    p = best.p();
    n = best.n();
  }
  h.p[0] = p.x; h.p[1] = p.y; h.p[2] = p.z;
  h.n[0] = n.x; h.n[1] = n.y; h.n[2] = n.z;
  hits[g] = h;
  if constexpr (UV) {
    if (best.prim >= 0) {
      DHitUV x;
      const d3 t = best.tan();
      x.u = best.u();
      x.v = best.v();
      x.tan[0] = t.x; x.tan[1] = t.y; x.tan[2] = t.z;
      x.pad = 0;
      hits_uv[g] = x;
    }
  }
}
template <bool UV> DEV void hit_load(const DHit *hits, const DHitUV *hits_uv, size_t g, Best<UV> &best) {
  const DHit h = hits[g];
  best.prim = h.prim;
  best.t = h.t;
  best.set_p(ld3(h.p));
  best.set_n(ld3(h.n));
  if constexpr (UV) {
    if (h.prim >= 0) {
      const DHitUV x = hits_uv[g];
      best.set_uv(x.u, x.v);
      best.set_tan(ld3(x.tan));
    }
  }
}

template <bool PRIMARY, bool UV, bool MOTION>
__global__ __launch_bounds__(256) WAVES_PER_SIMD(MIRT_WIDE_WALK_WAVES) void k_closest_stream(const TraceArgs a) {
  const DScene &sc = a.sc;
  unsigned n_node = 0, n_prim = 0;
  QueueRun run;
  if (PRIMARY) {
    run.begin = run.slen = 0;
    run.total = a.n * (uint32_t)a.fr.spp * 256u;  // (pixel block, sample, lane of the block)
  } else {
    run.open(a.in_count, a.in.shard_cap);
  }
  if (PRIMARY && a.progress && threadIdx.x == 0) atomicMin(a.progress, (unsigned long long)wall_clock64());
  WorkDealer dealer;
  bool has_ray = false;
  uint32_t mine = 0;  // the number of the ray in hand
  RayL ray = make_ray(mk(0, 0, 0), mk(0, 0, 1), 0.5);
  double closest = 0.0;
  Best<UV> best;
  BEST_IN_LDS(UV, best);
  WideWalk w;
  int32_t deep[WIDE_STACK_DEEP];
  w.start(ray, false);
  for (;;) {
    const bool found = has_ray && !w.busy();  // (or nothing to find)
    if (sc.tie_hi > 1.0 && __any(found && w.tied()))  // (no band: no tree in the reference's order to walk)
      walk_again_in_order<UV, MOTION>(sc, ray, found && w.tied(), 1e-5, 1.7976931348623157e308, best, n_node, n_prim);
    if (found) {  // the hit goes out, the lane is free
      hit_store<UV>(a.hits, a.hits_uv, mine, best);
      has_ray = false;
    }
    const bool want = !has_ray;
    if (!dealer.exhausted && ((int)__popcll(__ballot(want)) >= REFILL_MIN || !__any(has_ray))) {
      uint32_t g;
      const bool take = dealer.deal(a.grab, run.total, want, g);
      RaySeed r;
      bool ok = take;
      if (PRIMARY) {
        const bool inside = primary_ray(a, g >> 8, g & 255u, r);
        ok = take && inside;  // numbers that fall outside the image get no hit record; k_shade skips them too
      } else {
        queued_ray(a.in, run.slot(g, a.in.shard_cap), r);  // (lanes that took nothing read entry 0 and drop it)
      }
      if (ok) {
        ray = make_ray(r.ro, r.rd, r.time);
        closest = 1.7976931348623157e308;  // numeric_limits<double>::max() (Raytracer.cpp:300)
        best.prim = -1;
        best.t = 0;
        mine = g;
        has_ray = true;
        w.start(ray, true);
      }
    }
    if (!__any(has_ray)) {
      if (dealer.exhausted) break;
      continue;
    }
    wide_step<false, UV, MOTION>(sc, ray, 1e-5, closest, best, w, deep, n_node, n_prim);
  }
  count_add(&my_counters(a.counters)->node_tests, n_node);
  count_add(&my_counters(a.counters)->prim_tests, n_prim);
}

// Ray number g = virtual block vb * 256 + lane, the numbering k_closest_stream used.
template <bool PRIMARY, bool UV>
__global__ __launch_bounds__(256) void k_shade(const TraceArgs a) {
  QueueRun run;
  if (PRIMARY) {
    run.begin = run.slen = 0;
    run.total = a.n * (uint32_t)a.fr.spp * 256u;
  } else {
    run.open(a.in_count, a.in.shard_cap);
  }
  const uint32_t n_vb = (run.total + 255u) >> 8;
  for (uint32_t vb = blockIdx.x; vb < n_vb; vb += gridDim.x) {
    const uint32_t g = vb * 256u + threadIdx.x;
    bool active = g < run.total;
    RaySeed r;
    if (PRIMARY) {
      const bool inside = primary_ray(a, vb, threadIdx.x, r);
      active = active && inside;
    } else {
      queued_ray(a.in, run.slot(active ? g : 0, a.in.shard_cap), r);
    }
    const RayL ray = make_ray(r.ro, r.rd, r.time);
    Best<UV> best;
    BEST_IN_LDS(UV, best);
    if (active) hit_load<UV>(a.hits, a.hits_uv, g, best);
    shade_hits<UV>(a, a.out, level_of(a), vb % QUEUE_SHARDS, ray, best, active, r.W, r.key, r.pixel);
    if (PRIMARY) progress_mark(a, vb, true);
  }
}


// Raytracer::compute_shadow (Raytracer.cpp:145-208) for one (hit, light) pair per lane: the form used
// when every light of the frame has a single stratum (no area light is sampled more than once).  Each lane
// is a small state machine -- stratum index, distance walked, attenuation -- and the wave runs
// closest-hit segments until every lane is done, so lanes with many strata or glass on the way do
// not hold the others idle between their segments.  Shadow rays are at shutter time 0.5 (Ray.h:15).
//
// While a segment is being traced, the lane's walk state that the traversal does not touch is
// parked in LDS and the parts of the work item that are only needed later (colour, pixel, random
// key, light) stay in memory: the traversal with its inlined intersectors sets the kernel's vector
// register peak, and every register freed there counts towards one more resident wave per SIMD.
template <int WALK, bool ANYHIT>
DEV void shadow_block_items(const ShadowArgs &a, const LightTable &lights, bool valid, size_t slot, unsigned &n_node,
                      unsigned &n_prim, unsigned &n_seg) {
  // per lane: the item (point, colour, key, pixel, light) and the walk state
  __shared__ double park_p[3][256], park_c[3][256], park_w[4][256];  // park_w: total att cur dist
  __shared__ uint64_t park_key[256];
  __shared__ uint32_t park_pixel[256], park_light[256];
  const DScene &sc = a.sc;
  const mirt_config &cfg = a.fr.cfg;
  const int tid = threadIdx.x;
#if MIRT_PROFILE == 2
  PROF_DECL;
#endif
  d3 P;
  {
    const DShadowItem it = a.items[slot];
    P = ld3(it.p);
    park_p[0][tid] = it.p[0];
    park_p[1][tid] = it.p[1];
    park_p[2][tid] = it.p[2];
    park_c[0][tid] = it.c[0];
    park_c[1][tid] = it.c[1];
    park_c[2][tid] = it.c[2];
    park_key[tid] = it.key;
    park_pixel[tid] = it.pixel;
    park_light[tid] = it.light;
  }
  double eps = cfg.ray_offset_epsilon;
  if (cfg.use_adaptive_epsilon) eps += len(P) * cfg.adaptive_epsilon_scale;

  int s = 0;
  double total = 0.0, att = 1.0, cur = 0.0, dist = 0.0;
  d3 dir = mk(0, 0, 1);
  bool busy = false;
  // moves to the next stratum that has a segment to walk; returns false when all strata are done
  auto next_stratum = [&]() {
    const uint32_t li = park_light[tid];
    const int side = light_side(lights, a.fr, li), n = side * side;
    if (s >= n) return false;
    const uint64_t key = park_key[tid];
    while (s < n) {
      const d3 to = stratum_point(lights, sc, a.fr, li, key, s, side) - P;
      dist = len(to);
      dir = to / dist;
      att = 1.0;
      cur = eps;
      if (cur < dist) return true;
      s++;  // nothing to walk: this stratum is unoccluded
    }
    return false;
  };
  if (valid) busy = next_stratum();
#if MIRT_PROFILE == 2
  PROF_MARK(0);
#endif

  while (__any(busy)) {
    const RayL ray = make_ray(P + dir * cur, dir, 0.5);
    double closest = dist - cur;
    Best<false> best;  // shadow segments record no hit point
    best.rec = nullptr;
    best.prim = -1;
    best.t = 0;
    if (busy) n_seg++;
    park_w[0][tid] = total;
    park_w[1][tid] = att;
    park_w[2][tid] = cur;
    park_w[3][tid] = dist;
    PARK_FENCE();
#if MIRT_PROFILE == 2
    PROF_MARK(2);
#endif
    if constexpr (ANYHIT && WALK == WALK_PACKET)
      traverse_packet<true, false, false, true>(args_again(a).sc, ray, busy, eps, closest, best, n_node, n_prim, dist - cur);
    else
      traverse<WALK, true, false, false>(args_again(a).sc, ray, busy, eps, closest, best, n_node, n_prim);
#if MIRT_PROFILE == 2
    PROF_MARK(1);
#endif
    PARK_FENCE();
    P = mk(park_p[0][tid], park_p[1][tid], park_p[2][tid]);
    total = park_w[0][tid];
    att = park_w[1][tid];
    cur = park_w[2][tid];
    dist = park_w[3][tid];
    dir = ray.d;
    if (busy) {
      bool done = true;
      if (best.prim >= 0 && best.t < (dist - cur)) {
        const double tr = sc.prim_transparency[best.prim];
        if (tr > 0.0) {
          const double opacity = 1.0 - tr;
          att *= (1.0 - opacity);
          cur += best.t + eps;
          done = (att < 0.01) || !(cur < dist);
        } else {
          att = 0.0;
        }
      }
      if (done) {
        total += (1.0 - att);
        s++;
        busy = next_stratum();
      }
    }
  }
  {
    const double shadow = valid ? total * light_inv_n(lights, a.fr, park_light[tid]) : 1.0;
    const bool lit = valid && !(shadow >= 1.0);  // Raytracer.cpp:264,271
    pixel_add_wave(a.accum, a.fr.plane, park_pixel[tid], mk(park_c[0][tid], park_c[1][tid], park_c[2][tid]) * (1.0 - shadow), lit);
  }
#if MIRT_PROFILE == 2
  PROF_MARK(3);
  PROF_FLUSH(my_counters(a.counters));
#endif
}

// shadow_block_items when the walk may stop at any opaque hit (DScene::shadow_any_hit) and no light of the scene
// is an area light: an item is then one segment towards the light's position, no draw, no strata, nothing chained
// behind a clear surface, and the item's contribution goes to the pixel whole or not at all.  What the general
// form carries for the other cases (stratum counters, attenuation, the distance walked so far, the item's key,
// the lights' strata tables) is not there to be kept in registers or parked.
template <int WALK>
DEV void shadow_block_point(const ShadowArgs &a, const LightTable &lights, bool valid, size_t slot, unsigned &n_node,
                            unsigned &n_prim, unsigned &n_seg) {
  __shared__ double park_c[3][256];
  __shared__ uint32_t park_pixel[256];
  const DScene &sc = a.sc;
  const mirt_config &cfg = a.fr.cfg;
  const int tid = threadIdx.x;
  const DShadowItem it = a.items[slot];
  const d3 P = ld3(it.p);
  park_c[0][tid] = it.c[0];
  park_c[1][tid] = it.c[1];
  park_c[2][tid] = it.c[2];
  park_pixel[tid] = it.pixel;
  double eps = cfg.ray_offset_epsilon;
  if (cfg.use_adaptive_epsilon) eps += len(P) * cfg.adaptive_epsilon_scale;
  const uint32_t li = it.light;
  const d3 L = li < LIGHT_TABLE_MAX ? mk(lights.loc[li][0], lights.loc[li][1], lights.loc[li][2]) : ld3(sc.lights[li].location);
  const d3 to = L - P;
  const double dist = len(to);
  const d3 dir = to / dist;
  const bool busy = valid && eps < dist;  // (a light nearer than the offset shadows nothing, Raytracer.cpp:178-180)
  const RayL ray = make_ray(P + dir * eps, dir, 0.5);
  const double limit = dist - eps;
  double closest = limit;
  Best<false> best;  // shadow segments record no hit point
  best.rec = nullptr;
  best.prim = -1;
  best.t = 0;
  if (busy) n_seg++;
  PARK_FENCE();
  traverse_packet<true, false, false, true>(args_again(a).sc, ray, busy, eps, closest, best, n_node, n_prim, limit);
  PARK_FENCE();
  const bool occluded = busy && best.prim >= 0 && best.t < limit;
  const ShadowArgs &ap = args_again(a);
  pixel_add_wave(ap.accum, ap.fr.plane, park_pixel[tid], mk(park_c[0][tid], park_c[1][tid], park_c[2][tid]), valid && !occluded);
}

// Raytracer::compute_shadow (Raytracer.cpp:145-208) for the 64 (hit, light) items of a wave, when some
// light of the frame has more than one stratum.
//
// An item has side x side strata (one for a point light), each a chain of closest-hit segments that
// goes on through transparent objects.  The unit of work handed to a lane is one stratum of one item:
// the wave numbers its items' strata 0 .. N - 1 and a lane that has finished a stratum takes the next
// number nobody has taken, between two segments.  (One lane per item, strata one after the other, kept
// 33 of 64 lanes busy on a scene of area lights seen through glass: the lanes whose strata cross more
// surfaces held the wave.)  A stratum's 1 - attenuation is added to its item's sum in LDS; the lane
// that finishes an item's last stratum adds contribution x (1 - shadow) to the pixel
// (Raytracer.cpp:264-271).  The strata of an item are therefore summed in the order they finish, not
// in index order: the terms are 0 or 1 unless glass attenuates, and then the sum can differ from the
// reference's in its last bit.  Shadow rays are at shutter time 0.5 (Ray.h:15).
//
// While a segment is being traced, the lane's walk state that the traversal does not touch is
// parked in LDS and the item (point, colour, pixel, random key, light) stays there throughout: the
// traversal with its inlined intersectors sets the kernel's vector register peak, and every register
// freed there counts towards one more resident wave per SIMD.
template <int WALK, bool ANYHIT>
DEV void shadow_block_strata(const ShadowArgs &a, const LightTable &lights, bool valid, size_t slot, unsigned &n_node,
                      unsigned &n_prim, unsigned &n_seg) {
  // per item (indexed by the lane that loaded it)
  __shared__ double park_p[3][256], park_c[3][256], park_sum[256];
  __shared__ uint64_t park_key[256];
  __shared__ uint32_t park_pixel[256], park_light[256], park_first[256];
  __shared__ int32_t park_left[256];
  // per lane: the walk state of the stratum in hand (att cur dist)
  __shared__ double park_w[3][256];
  const DScene &sc = a.sc;
  const mirt_config &cfg = a.fr.cfg;
  const int tid = threadIdx.x, lane = tid & 63, w0 = tid & ~63;
#if MIRT_PROFILE == 2
  PROF_DECL;
#endif
  int n_mine = 0;
  if (valid) {
    const DShadowItem it = a.items[slot];
    park_p[0][tid] = it.p[0];
    park_p[1][tid] = it.p[1];
    park_p[2][tid] = it.p[2];
    park_c[0][tid] = it.c[0];
    park_c[1][tid] = it.c[1];
    park_c[2][tid] = it.c[2];
    park_key[tid] = it.key;
    park_pixel[tid] = it.pixel;
    park_light[tid] = it.light;
    const int side = light_side(lights, a.fr, it.light);
    n_mine = side * side;
  }
  park_sum[tid] = 0.0;
  park_left[tid] = n_mine;
  // number the wave's strata: item j owns first[j] .. first[j] + n_j - 1
  int incl = n_mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d);
    if (lane >= d) incl += v;
  }
  park_first[tid] = (uint32_t)(incl - n_mine);
  const int N = __shfl(incl, 63);
  // every item the same power-of-two number of strata (the usual case): the item is a shift away
  const int n0 = __builtin_amdgcn_readfirstlane(n_mine);
  const bool uniform = n0 > 0 && (n0 & (n0 - 1)) == 0 && __ballot(n_mine != n0) == 0;
  const int shift = uniform ? __ffs(n0) - 1 : 0;
  const unsigned long long below = (1ull << lane) - 1ull;
  PARK_FENCE();

  // the stratum is over: its term goes to the item; the lane that brings an item's last term finishes it
  const bool single = uniform && n0 == 1;  // one stratum per item (point lights): its term is the item's sum
  auto stratum_done = [&](int it, double term) {
    if (!single && term != 0.0) atomicAdd(&park_sum[it], term);
    if (single || atomicSub(&park_left[it], 1) == 1) {
      const double shadow = (single ? term : park_sum[it]) * light_inv_n(lights, a.fr, park_light[it]);
      if (!(shadow >= 1.0))  // Raytracer.cpp:264,271
        pixel_add(a.accum, a.fr.plane, park_pixel[it], mk(park_c[0][it], park_c[1][it], park_c[2][it]) * (1.0 - shadow));
    }
  };

  int it = -1;  // the item (index into the parked arrays) whose stratum this lane walks; < 0: none
  int next = 0; // the first stratum number nobody has taken (wave-uniform)
  double att = 1.0, cur = 0.0, dist = 0.0, eps = 0.0;
  d3 dir = mk(0, 0, 1), P = mk(0, 0, 0);
#if MIRT_PROFILE == 2
  PROF_MARK(0);
#endif
  for (;;) {
    // ---- lanes without a stratum take the next numbers ----------------------------------------------
    const bool idle = it < 0;
    const unsigned long long im = __ballot(idle);
    if (im != 0 && next < N) {
      const int id = next + (int)__popcll(im & below);
      if (idle && id < N) {
        int j;
        if (uniform) {
          j = id >> shift;
        } else {
          j = 0;
#pragma unroll
          for (int step = 32; step; step >>= 1)
            if (park_first[w0 + j + step] <= (uint32_t)id) j += step;
        }
        it = w0 + j;
        const int st = id - (int)park_first[it];
        const uint32_t li = park_light[it];
        P = mk(park_p[0][it], park_p[1][it], park_p[2][it]);
        eps = cfg.ray_offset_epsilon;
        if (cfg.use_adaptive_epsilon) eps += len(P) * cfg.adaptive_epsilon_scale;
        const d3 to = stratum_point(lights, sc, a.fr, li, park_key[it], st, light_side(lights, a.fr, li)) - P;
        dist = len(to);
        dir = to / dist;
        att = 1.0;
        cur = eps;
        if (!(cur < dist)) {  // nothing to walk: this stratum is unoccluded
          stratum_done(it, 0.0);
          it = -1;
        }
      }
      next += (int)__popcll(im);
    }
    const bool busy = it >= 0;
    if (!__any(busy)) {
      if (next >= N) break;
      continue;
    }
    // ---- one segment -----------------------------------------------------------------------------
    const RayL ray = make_ray(P + dir * cur, dir, 0.5);
    double closest = dist - cur;
    Best<false> best;  // shadow segments record no hit point
    best.rec = nullptr;
    best.prim = -1;
    best.t = 0;
    if (busy) n_seg++;
    park_w[0][tid] = att;
    park_w[1][tid] = cur;
    park_w[2][tid] = dist;
    PARK_FENCE();
#if MIRT_PROFILE == 2
    PROF_MARK(2);
#endif
    if constexpr (ANYHIT && WALK == WALK_PACKET)
      traverse_packet<true, false, false, true>(args_again(a).sc, ray, busy, eps, closest, best, n_node, n_prim, dist - cur);
    else
      traverse<WALK, true, false, false>(args_again(a).sc, ray, busy, eps, closest, best, n_node, n_prim);
#if MIRT_PROFILE == 2
    PROF_MARK(1);
#endif
    PARK_FENCE();
    att = park_w[0][tid];
    cur = park_w[1][tid];
    dist = park_w[2][tid];
    dir = ray.d;
    if (busy) {
      P = mk(park_p[0][it], park_p[1][it], park_p[2][it]);
      bool done = true;
      if (best.prim >= 0 && best.t < (dist - cur)) {
        const double tr = sc.prim_transparency[best.prim];
        if (tr > 0.0) {
          const double opacity = 1.0 - tr;
          att *= (1.0 - opacity);
          cur += best.t + eps;
          done = (att < 0.01) || !(cur < dist);
        } else {
          att = 0.0;
        }
      }
      if (done) {
        stratum_done(it, 1.0 - att);
        it = -1;
      }
    }
  }
#if MIRT_PROFILE == 2
  PROF_MARK(3);
  PROF_FLUSH(my_counters(a.counters));
#endif
}

// Fixed-size grid striding over the virtual blocks of the shadow queue.
// STRATA: some light of the frame has more than one stratum.
// ANYHIT (packet walk only): see traverse_packet; a stratum is then one segment, never a chain.
// POINT (with ANYHIT, the packet walk and no strata only): no light of the scene is an area light
template <int WALK, bool STRATA, bool ANYHIT = false, bool POINT = false>
__global__ __launch_bounds__(256) WAVES_FOR_WALK(WALK, ANYHIT ? MIRT_SHADOW_WAVES : 3) void k_shadow(const ShadowArgs a) {
  unsigned n_node = 0, n_prim = 0, n_seg = 0;
  __shared__ LightTable lights;
  light_table_fill(lights, a.sc, a.fr);
  const WaveView v = wave_view(a.count, a.shard_cap);
  for (uint32_t vb = blockIdx.x; vb < v.total; vb += gridDim.x) {
    size_t slot;
    const bool valid = view_entry(v, vb, a.shard_cap, slot);
    if (POINT) shadow_block_point<WALK>(a, lights, valid, slot, n_node, n_prim, n_seg);
    else if (STRATA) shadow_block_strata<WALK, ANYHIT>(a, lights, valid, slot, n_node, n_prim, n_seg);
    else shadow_block_items<WALK, ANYHIT>(a, lights, valid, slot, n_node, n_prim, n_seg);
  }
#ifndef MIRT_PROFILE
  count_add(&my_counters(a.counters)->shadow_node_tests, n_node);
  count_add(&my_counters(a.counters)->shadow_prim_tests, n_prim);
  count_add(&my_counters(a.counters)->shadow_segments, n_seg);
#endif
}


// k_shadow for the wide walk, with lanes that take their next work item as soon as they have
// finished one.  In k_shadow a wave is done with its 64 items when the slowest is, and on a large
// scene the slowest is slow: shadow rays from deep inside a cloud of primitives visit hundreds of
// boxes, rays that start near its edge a handful (measured: 7 of 64 lanes busy per vector instruction).
// Here the grid is resident (a few blocks per CU); a wave takes items off the launch's queue in chunks
// (one atomic per WORK_CHUNK items) and hands them to its idle lanes whenever at least REFILL_MIN
// lanes are idle, between steps of the walk.  What a lane does with an item is shadow_block's state
// machine unchanged: strata, segments through transparent objects, attenuation, one pixel add.
//
// ANYHIT: every primitive of the scene is opaque or perfectly clear (transparency exactly 1, which leaves
// the attenuation of Raytracer.cpp:190-198 as it is).  A stratum is then shadowed if and only if some
// opaque primitive lies between the point and the light, so its one segment may end at the first such hit
// found, in any order, and clear primitives need no test.  (The reference starts a new segment an epsilon
// behind every clear surface and so skips whatever lies within two epsilons behind it; an opaque primitive
// thin enough to lie wholly inside such a gap is found here and not there.)
template <bool ANYHIT>
__global__ __launch_bounds__(256) WAVES_PER_SIMD(MIRT_WIDE_WALK_WAVES) void k_shadow_stream(const ShadowArgs a) {
  __shared__ double park_p[3][256], park_c[3][256], park_w[4][256];  // park_w: total att cur dist
  __shared__ uint64_t park_key[256];
  __shared__ uint32_t park_pixel[256], park_light[256];
  __shared__ int32_t park_s[256];
  __shared__ LightTable lights;
  const DScene &sc = a.sc;
  const mirt_config &cfg = a.fr.cfg;
  const int tid = threadIdx.x;
  unsigned n_node = 0, n_prim = 0, n_seg = 0;
  light_table_fill(lights, sc, a.fr);

  QueueRun run;
  run.open(a.count, a.shard_cap);
  WorkDealer dealer;
  bool has_item = false;
  RayL ray = make_ray(mk(0, 0, 0), mk(0, 0, 1), 0.5);
  double eps = 0.0, closest = 0.0, limit = 0.0;
  Best<false> best;
  best.rec = nullptr;
  best.prim = -1;
  best.t = 0;
  WideWalk w;
  int32_t deep[WIDE_STACK_DEEP];
  w.start(ray, false);

  // Moves the lane to the next stratum that has a segment to walk and starts that segment; false
  // when all strata are done.  Strata state lives in LDS (park_s, park_w).
  auto begin_segment = [&](d3 P, d3 dir, double cur, double dist) {
    ray = make_ray(P + dir * cur, dir, 0.5);
    closest = dist - cur;
    limit = closest;
    best.prim = -1;
    best.t = 0;
    n_seg++;
    w.start(ray, true);
  };
  auto next_stratum = [&]() {
    const uint32_t li = park_light[tid];
    const int side = light_side(lights, a.fr, li), n = side * side;
    int s = park_s[tid];
    if (s >= n) return false;
    const uint64_t key = park_key[tid];
    const d3 P = mk(park_p[0][tid], park_p[1][tid], park_p[2][tid]);
    while (s < n) {
      const d3 to = stratum_point(lights, sc, a.fr, li, key, s, side) - P;
      const double dist = len(to);
      const d3 dir = to / dist;
      if (eps < dist) {
        park_s[tid] = s;
        park_w[1][tid] = 1.0;  // att
        park_w[2][tid] = eps;  // cur
        park_w[3][tid] = dist;
        begin_segment(P, dir, eps, dist);
        return true;
      }
      s++;  // nothing to walk: this stratum is unoccluded
    }
    park_s[tid] = s;
    return false;
  };
  auto finish_item = [&]() {
    const double shadow = park_w[0][tid] * light_inv_n(lights, a.fr, park_light[tid]);
    if (!(shadow >= 1.0))  // Raytracer.cpp:264,271
      pixel_add(a.accum, a.fr.plane, park_pixel[tid], mk(park_c[0][tid], park_c[1][tid], park_c[2][tid]) * (1.0 - shadow));
    has_item = false;
  };

  for (;;) {
    // ---- hand idle lanes their next item -----------------------------------------------------
    const bool want = !has_item;
    if (!dealer.exhausted && ((int)__popcll(__ballot(want)) >= REFILL_MIN || !__any(has_item))) {
      uint32_t g;
      const bool take = dealer.deal(a.grab, run.total, want, g);
      const size_t at = run.slot(g, a.shard_cap);
      if (take) {
        const DShadowItem it = a.items[at];
        park_p[0][tid] = it.p[0];
        park_p[1][tid] = it.p[1];
        park_p[2][tid] = it.p[2];
        park_c[0][tid] = it.c[0];
        park_c[1][tid] = it.c[1];
        park_c[2][tid] = it.c[2];
        park_key[tid] = it.key;
        park_pixel[tid] = it.pixel;
        park_light[tid] = it.light;
        park_s[tid] = 0;
        park_w[0][tid] = 0.0;  // total
        eps = cfg.ray_offset_epsilon;
        if (cfg.use_adaptive_epsilon) eps += len(ld3(it.p)) * cfg.adaptive_epsilon_scale;
        has_item = true;
        if (!next_stratum()) finish_item();
      }
    }
    if (!__any(has_item)) {
      if (dealer.exhausted) break;
      continue;  // nothing in hand but the queue is not empty: refill (the condition above holds now)
    }
    // ---- one step of the walk ----------------------------------------------------------------------
    wide_step<true, false, false, ANYHIT>(sc, ray, eps, closest, best, w, deep, n_node, n_prim, limit);
    // ---- lanes whose segment is over ---------------------------------------------------------------
    if (has_item && !w.busy()) {
      double att = park_w[1][tid], cur = park_w[2][tid];
      const double dist = park_w[3][tid];
      bool done = true;
      if (best.prim >= 0 && best.t < (dist - cur)) {
        const double tr = sc.prim_transparency[best.prim];
        if (tr > 0.0) {
          const double opacity = 1.0 - tr;
          att *= (1.0 - opacity);
          cur += best.t + eps;
          done = (att < 0.01) || !(cur < dist);
        } else {
          att = 0.0;
        }
      }
      if (!done) {  // on through the transparent object: the next segment of the same stratum
        park_w[1][tid] = att;
        park_w[2][tid] = cur;
        begin_segment(mk(park_p[0][tid], park_p[1][tid], park_p[2][tid]), ray.d, cur, dist);
      } else {
        park_w[0][tid] += (1.0 - att);
        park_s[tid] += 1;
        if (!next_stratum()) finish_item();
      }
    }
  }
  count_add(&my_counters(a.counters)->shadow_node_tests, n_node);
  count_add(&my_counters(a.counters)->shadow_prim_tests, n_prim);
  count_add(&my_counters(a.counters)->shadow_segments, n_seg);
}


// Raytracer.cpp:544-568: mean, optional tone map, gamma on un-clamped values.  Leaves the
// accumulation buffer zeroed, which spares the next frame a fill launch.
__global__ __launch_bounds__(256) void k_resolve(const ResolveArgs a) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= a.n_pixels) return;
  double c[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double x = a.accum[(size_t)k * a.n_pixels + i] / a.spp;
    a.accum[(size_t)k * a.n_pixels + i] = 0.0;
    if (a.cfg.tone_mapping_mode == MIRT_TONEMAP_REINHARD) {
      x = x / (1.0 + x);
    } else if (a.cfg.tone_mapping_mode == MIRT_TONEMAP_EXPOSURE) {
      x *= a.cfg.exposure;
      x = mind(1.0, x);
    }
    if (a.cfg.enable_gamma_correction) x = pow(x, 1.0 / a.cfg.gamma);
    c[k] = x;
  }
  a.out[3 * (size_t)i] = c[0];
  a.out[3 * (size_t)i + 1] = c[1];
  a.out[3 * (size_t)i + 2] = c[2];
}


// Pixel (x, y) of the frame from the compact tile buffer of the member that rendered it (tile_owner,
// tile_slot in device_types.h).
__global__ __launch_bounds__(256) void k_deinterleave(const DeinterleaveArgs a) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= (size_t)a.width * a.height) return;
  const int x = (int)(i % a.width), y = (int)(i / a.width);
  const int tx = x / a.tile, ty = y / a.tile;
  const int r = tile_owner(tx, ty, a.tiles_x, a.world, a.stagger), slot = tile_slot(tx, ty, a.tiles_x, a.world, a.stagger);
  const double *src = a.gathered + a.offset[r] +
                      3 * ((size_t)slot * a.tile * a.tile + (size_t)(y % a.tile) * a.tile + (x % a.tile));
  a.frame[3 * i] = src[0];
  a.frame[3 * i + 1] = src[1];
  a.frame[3 * i + 2] = src[2];
}

// Closest hit of the pixel-centre camera rays at time 0.5 (pinhole): the parity hook.
template <int WALK>
__global__ __launch_bounds__(256) void k_trace_ids(const IdsArgs a) {
  const uint32_t bx = blockIdx.x % ((a.width + 15) / 16), by = blockIdx.x / ((a.width + 15) / 16);
  const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int gx = (int)(bx * 16 + (wv & 1) * 8 + (ln & 7)), gy = (int)(by * 16 + (wv >> 1) * 8 + (ln >> 3));
  const bool active = gx < a.width && gy < a.height;
  d3 ro, rd;
  camera_ray(a.cam, gx + 0.5, gy + 0.5, a.width, a.height, 0, ro, rd);
  const RayL ray = make_ray(ro, rd, 0.5);
  Best<true> best;
  BEST_IN_LDS(true, best);
  double closest = 1.7976931348623157e308;
  unsigned nn = 0, np = 0;
  traverse<WALK, false, true, false>(a.sc, ray, active, 1e-5, closest, best, nn, np);
  if (!active) return;
  const size_t i = (size_t)gy * a.width + gx;
  const bool hit = best.prim >= 0;
  a.prim_id[i] = best.prim;
  Surface<true> sf = {};
  if (hit) sf = make_surface<true>(a.sc, ray, best);
  if (a.t) a.t[i] = hit ? best.t : 0.0;
  if (a.normal) {
    a.normal[3 * i] = hit ? sf.n.x : 0.0;
    a.normal[3 * i + 1] = hit ? sf.n.y : 0.0;
    a.normal[3 * i + 2] = hit ? sf.n.z : 0.0;
  }
  if (a.uv) {
    a.uv[2 * i] = hit ? sf.u : 0.0;
    a.uv[2 * i + 1] = hit ? sf.v : 0.0;
  }
}

// Closest hit of caller-supplied rays (any origin, direction, shutter time, range): parity hook.
template <int WALK>
__global__ __launch_bounds__(256) void k_trace_rays(const RaysArgs a) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const bool active = i < a.n;
  const size_t j = active ? i : 0;
  const RayL ray = make_ray(ld3(a.origin + 3 * j), ld3(a.dir + 3 * j), a.time ? a.time[j] : 0.5);
  Best<false> best;
  BEST_IN_LDS(false, best);
  double closest = a.t_max ? a.t_max[j] : 1.7976931348623157e308;
  unsigned nn = 0, np = 0;
  traverse<WALK, false, false, true>(a.sc, ray, active, a.t_min, closest, best, nn, np);
  if (!active) return;
  const bool hit = best.prim >= 0;
  a.prim_id[i] = best.prim;
  if (a.t) a.t[i] = hit ? best.t : 0.0;
  if (a.normal) {
    Surface<false> sf = {};
    if (hit) sf = make_surface<false>(a.sc, ray, best);
    a.normal[3 * (size_t)i] = sf.n.x;
    a.normal[3 * (size_t)i + 1] = sf.n.y;
    a.normal[3 * (size_t)i + 2] = sf.n.z;
  }
}
