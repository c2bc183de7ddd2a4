// BVH construction on the GPU, for scenes large enough that the host build is what a user waits for.
//
// The reference builds its tree on the host by median splits (Code/Geometry/Bvh.cpp:194-242) and the
// engine keeps that tree as the default, because two of the reference's rules make results depend on
// the tree: a box of zero thickness is never hit (BoundingBox.h:27), and two hits at exactly the same
// distance are won by the one tested later or earlier depending on the kind (Bvh.cpp:312-332).  This
// builder makes ANOTHER tree over the same primitives, directly in the form the wide walk reads
// (DWide / DLeaf, device_types.h), so it is offered only for scenes without planes -- every other
// primitive's box has volume -- and only to the wide walk, which does not keep the reference's tie
// order anyway.  The closest hit of a ray does not depend on the tree otherwise.
//
//   k_build_boxes    world box of every primitive, as get_*_bounding_box computes it: the eight
//                    corners of the object-space box through object_to_world, for a moving primitive
//                    the union of the boxes at matrix_t0 and matrix_t1 (e.g. Sphere.cpp:82-103); and
//                    the bounds of the box centres
//   k_build_keys     63-bit Morton code of every centre
//   k_sort_*         stable LSD radix sort of (code, primitive), 8 bits a pass, one wave per block
//   k_build_tree     binary radix tree over the sorted codes (every internal node finds its own
//                    range and split by binary search on common prefixes; equal codes are told
//                    apart by their position)
//   k_build_fit      boxes bottom-up: a thread per leaf climbs, the second thread to reach a node
//                    unites its children's boxes and climbs on
//   k_build_emit     top-down, one launch per level of the wide tree: a wide node is a binary node
//                    with its children collapsed into four slots (float boxes rounded outwards); a
//                    subtree of at most leaf_max primitives becomes one DLeaf with its exact box
//   k_build_refresh  refit: after the primitives moved, boxes and near/far order codes of every wide
//                    node and leaf again from the re-fitted binary tree; topology and links stay
#pragma once

#include "geometry.h"

DEV unsigned long long order_bits(double x) {  // monotone: a < b  <=>  order_bits(a) < order_bits(b)
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
DEV double order_value(unsigned long long e) {
  const unsigned long long b = (e >> 63) ? (e & 0x7fffffffffffffffull) : ~e;
  return __longlong_as_double((long long)b);
}

// Box of `lo..hi` under the 3x4 row-major map m, corner by corner in the order of
// Transform::transform_bbox (the same sums in the same order, so the same doubles).
DEV void mapped_box(const double *m, const double *lo, const double *hi, bool first, double *out) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const double x = (i & 1) ? hi[0] : lo[0], y = (i & 2) ? hi[1] : lo[1], z = (i & 4) ? hi[2] : lo[2];
    const double w[3] = {m[0] * x + m[1] * y + m[2] * z + m[3], m[4] * x + m[5] * y + m[6] * z + m[7],
                         m[8] * x + m[9] * y + m[10] * z + m[11]};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (first && i == 0) {
        out[k] = w[k];
        out[3 + k] = w[k];
      } else {
        out[k] = (w[k] < out[k]) ? w[k] : out[k];
        out[3 + k] = (out[3 + k] < w[k]) ? w[k] : out[3 + k];
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_build_boxes(const BuildArgs a) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  double c[3] = {0, 0, 0};
  bool counted = false;
  if (i < a.n) {
    const DPrim &p = a.prims[i];
    double lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1};  // unit sphere, [-1,1]^3 cube (Sphere.cpp:83, Cube.cpp:151)
    if (p.type == MIRT_TORUS) {  // Torus.cpp:308-314
      const double R = p.param[0] + p.param[1], r = p.param[1];
      lo[0] = lo[1] = -R; hi[0] = hi[1] = R; lo[2] = -r; hi[2] = r;
    } else if (p.type == MIRT_CYLINDER || p.type == MIRT_CONE) {  // Cylinder.cpp:118-121, Cone.cpp:173-176
      const double r = p.param[0], h = p.param[1] / 2.0;
      lo[0] = lo[1] = -r; hi[0] = hi[1] = r; lo[2] = -h; hi[2] = h;
    }
    double b[6];
    if (p.motion != MOTION_NONE) {
      const double *m = a.motion_ends + 24 * (size_t)p.motion_index;
      mapped_box(m, lo, hi, true, b);
      mapped_box(m + 12, lo, hi, false, b);
    } else {
      mapped_box(p.xf + 12, lo, hi, true, b);
    }
    double *dst = a.pbox + 6 * (size_t)i;
#pragma unroll
    for (int k = 0; k < 6; k++) dst[k] = b[k];
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = (b[k] + b[3 + k]) * 0.5;
    counted = isfinite(c[0]) && isfinite(c[1]) && isfinite(c[2]);
  }
  // bounds of the centres: wave reduction, then one atomic per wave and word
  unsigned long long lo_e[3], hi_e[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    lo_e[k] = counted ? order_bits(c[k]) : ~0ull;
    hi_e[k] = counted ? order_bits(c[k]) : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long l2 = __shfl_xor(lo_e[k], off), h2 = __shfl_xor(hi_e[k], off);
      lo_e[k] = l2 < lo_e[k] ? l2 : lo_e[k];
      hi_e[k] = h2 > hi_e[k] ? h2 : hi_e[k];
    }
  }
  if (__lane_id() == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (lo_e[k] != ~0ull) atomicMin(a.bounds + k, lo_e[k]);
      if (hi_e[k] != 0ull) atomicMax(a.bounds + 3 + k, hi_e[k]);
    }
  }
}

DEV uint64_t spread21(uint64_t v) {  // bit i of v to bit 3 i
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

__global__ __launch_bounds__(256) void k_build_keys(const BuildArgs a) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= a.n) return;
  const double *b = a.pbox + 6 * (size_t)i;
  uint64_t code = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double lo = order_value(a.bounds[k]), hi = order_value(a.bounds[3 + k]);
    const double c = (b[k] + b[3 + k]) * 0.5, ext = hi - lo;
    double q = ext > 0.0 ? (c - lo) / ext * 2097152.0 : 0.0;
    q = q >= 0.0 ? q : 0.0;  // also a centre that is not a number
    q = q < 2097151.0 ? q : 2097151.0;
    code |= spread21((uint64_t)q) << k;
  }
  a.keys[i] = code;
  a.ids[i] = i;
}

// ---- radix sort: one 8-bit digit per pass, one wave per block, BUILD_SORT_TILE keys per block ----------
__global__ __launch_bounds__(64) void k_sort_hist(const BuildArgs a) {
  __shared__ uint32_t h[256];
  const uint32_t lane = threadIdx.x;
  for (int d = lane; d < 256; d += 64) h[d] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * BUILD_SORT_TILE;
  for (int r = 0; r < BUILD_SORT_ITEMS; r++) {
    const uint32_t i = base + r * 64 + lane;
    if (i < a.n) atomicAdd(&h[(uint32_t)(a.keys[i] >> a.shift) & 255u], 1u);
  }
  __syncthreads();
  for (int d = lane; d < 256; d += 64) a.hist[(size_t)d * a.sort_blocks + blockIdx.x] = h[d];
}

// exclusive prefix sum over the 256 x sort_blocks counts, digit-major: where each block's keys of each
// digit start in the output.  One block of 16 waves; wave w owns a contiguous sixteenth of the counts and goes
// through it 64 at a time (one coalesced load, a shuffle scan, a running carry): once to total it, and --
// after the 16 totals have been scanned -- once more to write the prefixes.
__global__ __launch_bounds__(1024) void k_sort_scan(const BuildArgs a) {
  __shared__ uint32_t part[16];
  const uint32_t total = 256u * a.sort_blocks;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t per = ((total + 15u) / 16u + 63u) & ~63u;  // counts per wave, a multiple of 64
  const uint32_t lo = wave * per, hi = lo + per < total ? lo + per : total;
  uint32_t s = 0;
  for (uint32_t i = lo + lane; i < hi; i += 64) s += a.hist[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) part[wave] = s;
  __syncthreads();
  uint32_t carry = 0;
  for (uint32_t w = 0; w < wave; w++) carry += part[w];
  for (uint32_t base = lo; base < hi; base += 64) {
    const uint32_t i = base + lane;
    const uint32_t c = i < hi ? a.hist[i] : 0;
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(incl, off);
      if (lane >= (uint32_t)off) incl += v;
    }
    if (i < hi) a.hist[i] = carry + incl - c;
    carry += __shfl(incl, 63);
  }
}

__global__ __launch_bounds__(64) void k_sort_scatter(const BuildArgs a) {
  __shared__ uint32_t off[256];
  const uint32_t lane = threadIdx.x;
  for (int d = lane; d < 256; d += 64) off[d] = a.hist[(size_t)d * a.sort_blocks + blockIdx.x];
  __syncthreads();
  const uint32_t base = blockIdx.x * BUILD_SORT_TILE;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int r = 0; r < BUILD_SORT_ITEMS; r++) {
    const uint32_t i = base + r * 64 + lane;
    const bool valid = i < a.n;
    const uint64_t key = valid ? a.keys[i] : 0;
    const uint32_t id = valid ? a.ids[i] : 0;
    const uint32_t digit = (uint32_t)(key >> a.shift) & 255u;
    // the lanes of the wave that hold the same digit
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (digit >> b) & 1u;
      const unsigned long long m = __ballot(valid && bit);
      same &= bit ? m : ~m;
    }
    const uint32_t rank = (uint32_t)__popcll(same & below);
    uint32_t start = 0;
    if (valid && rank == 0) {  // the first of them moves the digit's cursor for all
      start = off[digit];
      off[digit] = start + (uint32_t)__popcll(same);
    }
    const int leader = valid ? __ffsll((long long)same) - 1 : 0;
    start = __shfl(start, leader);
    if (valid) {
      a.keys_alt[start + rank] = key;
      a.ids_alt[start + rank] = id;
    }
    __syncthreads();  // the cursors are read again in the next round
  }
}

// ---- binary radix tree over the sorted codes ---------------------------------------------------------
// Length of the common prefix of the codes at sorted positions i and j (-1 when j is outside); equal
// codes go on with their positions, so that all n "keys" are distinct.
DEV int common_prefix(const uint64_t *keys, uint32_t n, long long i, long long j) {
  if (j < 0 || j >= (long long)n) return -1;
  const uint64_t x = keys[i] ^ keys[j];
  if (x != 0) return __clzll((long long)x);
  return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

__global__ __launch_bounds__(256) void k_build_tree(const BuildArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const uint32_t n = a.n;
  if (i >= (long long)n - 1) return;
  const uint64_t *keys = a.keys;
  const int d = common_prefix(keys, n, i, i + 1) > common_prefix(keys, n, i, i - 1) ? 1 : -1;
  const int dmin = common_prefix(keys, n, i, i - d);
  long long lmax = 2;
  while (common_prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  long long l = 0;
  for (long long t = lmax / 2; t >= 1; t /= 2)
    if (common_prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const long long j = i + l * d;
  const int dnode = common_prefix(keys, n, i, j);
  long long s = 0, t = l;
  do {
    t = (t + 1) / 2;
    if (common_prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const long long gamma = i + s * d + (d < 0 ? -1 : 0);
  const long long first = i < j ? i : j, last = i < j ? j : i;
  const int32_t left = first == gamma ? ~(int32_t)gamma : (int32_t)gamma;
  const int32_t right = last == gamma + 1 ? ~(int32_t)(gamma + 1) : (int32_t)(gamma + 1);
  a.child[2 * i] = left;
  a.child[2 * i + 1] = right;
  a.range[2 * i] = (uint32_t)first;
  a.range[2 * i + 1] = (uint32_t)last;
  a.parent[left < 0 ? (n - 1) + (uint32_t)~left : (uint32_t)left] = (int32_t)i;
  a.parent[right < 0 ? (n - 1) + (uint32_t)~right : (uint32_t)right] = (int32_t)i;
  if (i == 0) a.parent[0] = -1;
}

// box of a node reference, read past the caches' stale lines (written by other threads of the same launch)
DEV void ref_box(const BuildArgs &a, int32_t x, double *out) {
  const volatile double *src = x < 0 ? a.pbox + 6 * (size_t)a.ids[~x] : a.nbox + 6 * (size_t)x;
#pragma unroll
  for (int k = 0; k < 6; k++) out[k] = src[k];
}

__global__ __launch_bounds__(256) void k_build_fit(const BuildArgs a) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= a.n) return;
  int32_t node = a.parent[(a.n - 1) + k];
  while (node >= 0) {
    __threadfence();  // what this thread wrote below is visible before it reports in
    if (atomicAdd(a.visits + node, 1u) == 0) return;  // the other child's thread will fit this node
    __threadfence();
    double l[6], r[6];
    ref_box(a, a.child[2 * (size_t)node], l);
    ref_box(a, a.child[2 * (size_t)node + 1], r);
    double *dst = a.nbox + 6 * (size_t)node;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      dst[c] = (r[c] < l[c]) ? r[c] : l[c];
      dst[3 + c] = (l[3 + c] < r[3 + c]) ? r[3 + c] : l[3 + c];
    }
    node = a.parent[node];
  }
}

// ---- emission of the wide tree -----------------------------------------------------------------------
DEV float float_below(double x) {  // the largest float <= x
  float f = (float)x;
  if ((double)f > x) {
    const uint32_t b = __float_as_uint(f);
    f = f > 0.0f ? __uint_as_float(b - 1u) : (f < 0.0f ? __uint_as_float(b + 1u) : __uint_as_float(0x80000001u));
  }
  return f;
}
DEV float float_above(double x) {  // the smallest float >= x
  float f = (float)x;
  if ((double)f < x) {
    const uint32_t b = __float_as_uint(f);
    f = f > 0.0f ? __uint_as_float(b + 1u) : (f < 0.0f ? __uint_as_float(b - 1u) : __uint_as_float(0x00000001u));
  }
  return f;
}

DEV bool ref_is_leaf(const BuildArgs &a, int32_t x) {
  return x < 0 || a.range[2 * (size_t)x + 1] - a.range[2 * (size_t)x] + 1u <= a.leaf_max;
}
// near/far code of a binary inner node, as the ordered walks read it: the axis along which the children's
// boxes lie furthest apart, and whether the second child is the lower one along it
DEV uint32_t ref_order(const BuildArgs &a, int32_t x) {
  double l[6], r[6];
  ref_box(a, a.child[2 * (size_t)x], l);
  ref_box(a, a.child[2 * (size_t)x + 1], r);
  int axis = 0;
  double best = -1.0;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const double sep = fabs((r[c] + r[3 + c]) - (l[c] + l[3 + c]));
    if (sep > best) {
      best = sep;
      axis = c;
    }
  }
  const bool second_lower = (r[axis] + r[3 + axis]) < (l[axis] + l[3 + axis]);
  return (uint32_t)axis | (second_lower ? 4u : 0u);
}
// The four slots of the wide node of binary node t: its children's children; a child that is a leaf
// fills the first slot of its pair alone.  A tree that is a single leaf has it in slot 0.
DEV uint32_t wide_slots(const BuildArgs &a, int32_t t, int32_t *slot) {
#pragma unroll
  for (int c = 0; c < 4; c++) slot[c] = BUILD_REF_NONE;
  if (ref_is_leaf(a, t)) {
    slot[0] = t;
    return 0;
  }
  uint32_t order = ref_order(a, t);
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const int32_t x = a.child[2 * (size_t)t + g];
    if (ref_is_leaf(a, x)) {
      slot[2 * g] = x;
    } else {
      order |= ref_order(a, x) << (3 * (g + 1));
      slot[2 * g] = a.child[2 * (size_t)x];
      slot[2 * g + 1] = a.child[2 * (size_t)x + 1];
    }
  }
  return order;
}
DEV void wide_boxes(const BuildArgs &a, const int32_t *slot, DWide &w) {
#pragma unroll
  for (int c = 0; c < 4; c++) {
    if (slot[c] == BUILD_REF_NONE) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        w.lo[c][k] = __builtin_huge_valf();
        w.hi[c][k] = -__builtin_huge_valf();
      }
    } else {
      double b[6];
      ref_box(a, slot[c], b);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        w.lo[c][k] = float_below(b[k]);
        w.hi[c][k] = float_above(b[3 + k]);
      }
    }
  }
}
DEV void leaf_box(const BuildArgs &a, int32_t x, DLeaf &l) {
  double b[6];
  ref_box(a, x, b);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    l.bmin[k] = b[k];
    l.bmax[k] = b[3 + k];
  }
}

__global__ __launch_bounds__(256) void k_build_emit(const BuildArgs a) {
  const uint32_t me = a.level_begin + blockIdx.x * 256u + threadIdx.x;
  if (me >= a.level_end) return;
  const int32_t t = a.wide_src[me];
  int32_t slot[4];
  DWide w;
  w.order = wide_slots(a, t, slot);
  w.pad[0] = w.pad[1] = w.pad[2] = 0;
  wide_boxes(a, slot, w);
  int inner = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) inner += (slot[c] != BUILD_REF_NONE && !ref_is_leaf(a, slot[c])) ? 1 : 0;
  uint32_t next = inner ? atomicAdd(a.counters, (uint32_t)inner) : 0;  // the children's wide nodes, side by side
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int32_t x = slot[c];
    if (x == BUILD_REF_NONE) {
      w.child[c] = WIDE_NONE;
    } else if (!ref_is_leaf(a, x)) {
      a.wide_src[next] = x;
      w.child[c] = (int32_t)next++;
    } else {
      const uint32_t li = atomicAdd(a.counters + 1, 1u);
      DLeaf l;
      leaf_box(a, x, l);
      const uint32_t first = x < 0 ? (uint32_t)~x : a.range[2 * (size_t)x];
      const uint32_t count = x < 0 ? 1u : a.range[2 * (size_t)x + 1] - first + 1u;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        int32_t tag = -1;
        if (k < count) {
          const uint32_t id = a.ids[first + k];
          tag = (int32_t)(id | ((uint32_t)a.prims[id].type << LEAF_KIND_SHIFT) |
                          (a.transparency[id] == 1.0 ? (uint32_t)LEAF_CLEAR_BIT : 0u));
        }
        l.prim[k] = tag;
      }
      a.leaves[li] = l;
      a.leaf_src[li] = x;
      w.child[c] = ~(int32_t)li;
    }
  }
  a.wide[me] = w;
}

// Refit: the binary tree's boxes have been fitted again (k_build_boxes, k_build_fit); bring the wide
// nodes' float boxes and order codes and the leaves' exact boxes up to date.  Links stay.
__global__ __launch_bounds__(256) void k_build_refresh(const BuildArgs a, uint32_t n_wide, uint32_t n_leaves) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n_wide) {
    int32_t slot[4];
    DWide w = a.wide[i];
    w.order = wide_slots(a, a.wide_src[i], slot);
    wide_boxes(a, slot, w);
    a.wide[i] = w;
  }
  if (i < n_leaves) {
    DLeaf l = a.leaves[i];
    leaf_box(a, a.leaf_src[i], l);
    a.leaves[i] = l;
  }
}

// ---- launchers -----------------------------------------------------------------------------------------
static unsigned blocks_of(uint32_t n, uint32_t per) { return n ? (n + per - 1) / per : 1; }
void launch_build_boxes(void *stream, const BuildArgs &a) {
  hipLaunchKernelGGL(k_build_boxes, dim3(blocks_of(a.n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
}
void launch_build_keys(void *stream, const BuildArgs &a) {
  hipLaunchKernelGGL(k_build_keys, dim3(blocks_of(a.n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
}
void launch_build_sort_pass(void *stream, const BuildArgs &a) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_sort_hist, dim3(a.sort_blocks), dim3(64), 0, st, a);
  hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(k_sort_scatter, dim3(a.sort_blocks), dim3(64), 0, st, a);
}
void launch_build_tree(void *stream, const BuildArgs &a) {
  hipLaunchKernelGGL(k_build_tree, dim3(blocks_of(a.n > 0 ? a.n - 1 : 0, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
}
void launch_build_fit(void *stream, const BuildArgs &a) {
  hipLaunchKernelGGL(k_build_fit, dim3(blocks_of(a.n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
}
void launch_build_emit(void *stream, const BuildArgs &a) {
  hipLaunchKernelGGL(k_build_emit, dim3(blocks_of(a.level_end - a.level_begin, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
}
void launch_build_refresh(void *stream, const BuildArgs &a, uint32_t n_wide, uint32_t n_leaves) {
  const uint32_t m = n_wide > n_leaves ? n_wide : n_leaves;
  hipLaunchKernelGGL(k_build_refresh, dim3(blocks_of(m, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                     n_wide, n_leaves);
}
