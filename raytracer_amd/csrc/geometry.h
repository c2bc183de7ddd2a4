// Device geometry: 3-vectors, the three kinds of object<->world map, and the six ray/primitive
// intersectors.  The intersectors follow the reference's tests, tolerances and tie rules
// (Code/Geometry/{Sphere,Cube,Plane,Torus,Cylinder,Cone}.cpp) so that the same object wins the same
// pixel; what differs is how the transform reaches them (see Xf* below) and that texture
// coordinates and the tangent frame are only computed in the kernels compiled for scenes that
// sample an image map.
#pragma once

#include <hip/hip_runtime.h>

#include "device_types.h"

#ifndef DEV
#define DEV __device__ __forceinline__
#endif

// Scene records are read through the constant address space: the kernels never write them, and a
// load at a wave-uniform constant address is always issued on the scalar data path (one fetch per
// wave, operands in scalar registers), whatever stores the kernel has made before it.
#define RO __attribute__((address_space(4)))
typedef const RO double *ro_doubles;
template <class T> DEV const RO T *rdonly(const T *p) { return (const RO T *)p; }

struct d3 {
  double x, y, z;
};
DEV d3 mk(double x, double y, double z) { return d3{x, y, z}; }
DEV d3 ld3(const double *p) { return d3{p[0], p[1], p[2]}; }
DEV d3 ld3(ro_doubles p) { return d3{p[0], p[1], p[2]}; }
DEV d3 operator+(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV d3 operator*(d3 a, d3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
DEV d3 operator*(d3 a, double c) { return {a.x * c, a.y * c, a.z * c}; }
DEV d3 operator/(d3 a, double c) { return {a.x / c, a.y / c, a.z / c}; }
DEV d3 operator-(d3 a) { return {-a.x, -a.y, -a.z}; }
DEV double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV d3 cross(d3 a, d3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DEV double len2(d3 a) { return dot(a, a); }
DEV double len(d3 a) { return sqrt(len2(a)); }
DEV d3 unit(d3 a) { return a / len(a); }
DEV double comp(d3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// A ray as the traversal sees it.
struct RayL {
  d3 o, d;
  d3 inv;       // 1 / d, per axis (BoundingBox.h:17)
  double time;
  double dlen;  // |d| (the reference divides hit distances by it, e.g. Sphere.cpp:46)
};
DEV RayL make_ray(d3 o, d3 d, double time) {
  RayL r;
  r.o = o;
  r.d = d;
  r.inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
  r.time = time;
  r.dlen = len(d);
  return r;
}

// Closest hit so far of one ray.  Distance and primitive stay in registers; what is only read
// when the traversal is over (point, normal, texture coordinates, tangent) is written straight to
// a per-lane record in LDS, so that it does not occupy vector registers while the intersectors run.
#define LDS_AS __attribute__((address_space(3)))
#define HIT_STRIDE 256  // doubles between two fields of a lane's record = threads per block
#define HIT_FIELDS(UV) ((UV) ? 11 : 6)
template <bool UV> struct Best {
  double t;
  int prim;            // -1: none
  LDS_AS double *rec;  // field k at rec[k * HIT_STRIDE]: p 0-2, n 3-5, u 6, v 7, tan 8-10
  DEV void put3(int k, d3 v) const {
    rec[k * HIT_STRIDE] = v.x;
    rec[(k + 1) * HIT_STRIDE] = v.y;
    rec[(k + 2) * HIT_STRIDE] = v.z;
  }
  DEV d3 get3(int k) const { return d3{rec[k * HIT_STRIDE], rec[(k + 1) * HIT_STRIDE], rec[(k + 2) * HIT_STRIDE]}; }
  // world hit point; outward world normal before normalisation (planes: already unit)
  DEV void set_p(d3 v) const { put3(0, v); }
  DEV void set_n(d3 v) const { put3(3, v); }
  DEV d3 p() const { return get3(0); }
  DEV d3 n() const { return get3(3); }
  // texture coordinates; object-space tangent mapped to world, un-normalised (planes: the unit edge direction)
  DEV void set_uv(double u, double v) const {
    rec[6 * HIT_STRIDE] = u;
    rec[7 * HIT_STRIDE] = v;
  }
  DEV void set_tan(d3 v) const { put3(8, v); }
  DEV double u() const { return rec[6 * HIT_STRIDE]; }
  DEV double v() const { return rec[7 * HIT_STRIDE]; }
  DEV d3 tan() const { return get3(8); }
};
// Declares the block's hit records and binds `best` to this lane's.
#define BEST_IN_LDS(UV, best)                              \
  __shared__ double hit_records[HIT_FIELDS(UV)][HIT_STRIDE]; \
  best.rec = (LDS_AS double *)&hit_records[0][threadIdx.x];  \
  best.prim = -1;                                          \
  best.t = 0

// ---- object <-> world maps --------------------------------------------------------------------
// XfS: the static 3x4 pair stored in DPrim::xf.  The pointer is wave-uniform in the packet
//      traversal, so the 24 values live in scalar registers.
struct XfS {
  ro_doubles m;
  DEV d3 obj_point(d3 p) const {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
  }
  DEV d3 obj_dir(d3 d) const {
    return {m[0] * d.x + m[1] * d.y + m[2] * d.z, m[4] * d.x + m[5] * d.y + m[6] * d.z,
            m[8] * d.x + m[9] * d.y + m[10] * d.z};
  }
  DEV d3 world_point(d3 p) const {
    return {m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15], m[16] * p.x + m[17] * p.y + m[18] * p.z + m[19],
            m[20] * p.x + m[21] * p.y + m[22] * p.z + m[23]};
  }
  DEV d3 world_dir(d3 d) const {
    return {m[12] * d.x + m[13] * d.y + m[14] * d.z, m[16] * d.x + m[17] * d.y + m[18] * d.z,
            m[20] * d.x + m[21] * d.y + m[22] * d.z};
  }
  // world_to_object transposed times n (Transform.h:61-67), not normalised
  DEV d3 world_normal(d3 n) const {
    return {m[0] * n.x + m[4] * n.y + m[8] * n.z, m[1] * n.x + m[5] * n.y + m[9] * n.z,
            m[2] * n.x + m[6] * n.y + m[10] * n.z};
  }
};

// XfT: constant rotation*scale block, translation interpolated per ray.
struct XfT {
  ro_doubles m;        // DMotion::m (9) followed by DMotion::minv (9)
  d3 t;
  DEV d3 obj_dir(d3 d) const {
    ro_doubles i = m + 9;
    return {i[0] * d.x + i[1] * d.y + i[2] * d.z, i[3] * d.x + i[4] * d.y + i[5] * d.z,
            i[6] * d.x + i[7] * d.y + i[8] * d.z};
  }
  DEV d3 obj_point(d3 p) const { return obj_dir(p - t); }
  DEV d3 world_dir(d3 d) const {
    return {m[0] * d.x + m[1] * d.y + m[2] * d.z, m[3] * d.x + m[4] * d.y + m[5] * d.z,
            m[6] * d.x + m[7] * d.y + m[8] * d.z};
  }
  DEV d3 world_point(d3 p) const { return world_dir(p) + t; }
  DEV d3 world_normal(d3 n) const {
    ro_doubles i = m + 9;
    return {i[0] * n.x + i[3] * n.y + i[6] * n.z, i[1] * n.x + i[4] * n.y + i[7] * n.z,
            i[2] * n.x + i[5] * n.y + i[8] * n.z};
  }
};

// XfG: translation, rotation and scale all interpolated per ray: o2w = T R S, w2o = S^-1 R^T T^-1.
struct XfG {
  double r[9];
  d3 s, is, t;
  DEV d3 rt(d3 d) const {  // R^T d
    return {r[0] * d.x + r[3] * d.y + r[6] * d.z, r[1] * d.x + r[4] * d.y + r[7] * d.z,
            r[2] * d.x + r[5] * d.y + r[8] * d.z};
  }
  DEV d3 rr(d3 d) const {  // R d
    return {r[0] * d.x + r[1] * d.y + r[2] * d.z, r[3] * d.x + r[4] * d.y + r[5] * d.z,
            r[6] * d.x + r[7] * d.y + r[8] * d.z};
  }
  DEV d3 obj_dir(d3 d) const { return rt(d) * is; }
  DEV d3 obj_point(d3 p) const { return rt(p - t) * is; }
  DEV d3 world_dir(d3 d) const { return rr(d * s); }
  DEV d3 world_point(d3 p) const { return rr(p * s) + t; }
  DEV d3 world_normal(d3 n) const { return rr(n * is); }
};

// sin and cos of an angle known to lie within a few turns of zero (|x| < 100), for the slerp angle in [0, pi] of
// a moving primitive.  The general routine carries an argument reduction for angles up to 1e308 that this never
// needs; here the quarter turn count is small, so k * (the leading 33 bits of pi/2) is exact and two
// subtractions reduce the angle to [-pi/4, pi/4] with an absolute error of about 1e-20, and the two polynomials
// are the usual minimax ones for that interval (error below one unit in the last place of the result).
// Not for directions the reference computes with sin and cos themselves (lobe, lens and light samples): those
// have to come out bit for bit, and the library's do; the interpolated transform is already composed by other
// formulas than the reference's, equal to about 1e-16.
DEV void sincos_bounded(double x, double *sn, double *cs) {
  const double k = rint(x * 0.63661977236758134308);  // 2 / pi
  const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
  const double z = r * r;
  double ps = 1.58969099521155010221e-10;
  ps = ps * z - 2.50507602534068634195e-08;
  ps = ps * z + 2.75573137070700676789e-06;
  ps = ps * z - 1.98412698298579493134e-04;
  ps = ps * z + 8.33333333332248946124e-03;
  ps = ps * z - 1.66666666666666324348e-01;
  const double s = r + (r * z) * ps;
  double pc = -1.13596475577881948265e-11;
  pc = pc * z + 2.08757232129817482790e-09;
  pc = pc * z - 2.75573143513906633035e-07;
  pc = pc * z + 2.48015872894767294178e-05;
  pc = pc * z - 1.38888888888741095749e-03;
  pc = pc * z + 4.16666666666666019037e-02;
  const double c = (1.0 - 0.5 * z) + (z * z) * pc;
  const int q = (int)k;
  const double a = (q & 1) ? c : s, b = (q & 1) ? s : c;
  *sn = (q & 2) ? -a : a;
  *cs = ((q + 1) & 2) ? -b : b;
}

// Mat4::interpolate (Transform.cpp:239-256) with the decomposition already done: lerp T and S,
// slerp or normalised lerp R (Quaternion.h:73-116), compose (Transform.cpp:209-236).
DEV XfG motion_general(const RO DMotion &mo, double t) {
  XfG x;
  x.t = mk(mo.t0[0] + mo.dt[0] * t, mo.t0[1] + mo.dt[1] * t, mo.t0[2] + mo.dt[2] * t);
  x.s = mk(mo.s0[0] + mo.ds[0] * t, mo.s0[1] + mo.ds[1] * t, mo.s0[2] + mo.ds[2] * t);
  x.is = mk(1.0 / x.s.x, 1.0 / x.s.y, 1.0 / x.s.z);
  double qw, qx, qy, qz;
  if (mo.lerp) {
    qw = mo.q0[0] + mo.dq[0] * t;
    qx = mo.q0[1] + mo.dq[1] * t;
    qy = mo.q0[2] + mo.dq[2] * t;
    qz = mo.q0[3] + mo.dq[3] * t;
    const double l = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    if (l < 1e-10) {
      qw = 1.0; qx = 0.0; qy = 0.0; qz = 0.0;
    } else {
      qw /= l; qx /= l; qy /= l; qz /= l;
    }
  } else {
    const double th = mo.theta0 * t;
    double sn, cs;
    sincos_bounded(th, &sn, &cs);
    const double k1 = sn * mo.inv_sin0;
    const double k0 = cs - mo.dot * k1;
    qw = k0 * mo.q0[0] + k1 * mo.dq[0];
    qx = k0 * mo.q0[1] + k1 * mo.dq[1];
    qy = k0 * mo.q0[2] + k1 * mo.dq[2];
    qz = k0 * mo.q0[3] + k1 * mo.dq[3];
  }
  const double xx = qx * qx, yy = qy * qy, zz = qz * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz;
  const double wx = qw * qx, wy = qw * qy, wz = qw * qz;
  x.r[0] = 1.0 - 2.0 * (yy + zz); x.r[1] = 2.0 * (xy - wz);       x.r[2] = 2.0 * (xz + wy);
  x.r[3] = 2.0 * (xy + wz);       x.r[4] = 1.0 - 2.0 * (xx + zz); x.r[5] = 2.0 * (yz - wx);
  x.r[6] = 2.0 * (xz - wy);       x.r[7] = 2.0 * (yz + wx);       x.r[8] = 1.0 - 2.0 * (xx + yy);
  return x;
}

DEV bool same(double a, double b) { return fabs(a - b) <= 2.220446049250313e-16; }  // Intersections.cpp:6-8

// Fills the texture coordinates and the tangent from the object-space tangent.  Normalisation and
// Gram-Schmidt against the final normal happen once, for the winning hit; the reference overwrites
// the transformed bitangent with normal x tangent (e.g. Sphere.cpp:72-77), so it is never needed.
template <class X> DEV void set_frame(Best<false> &, const X &, d3, d3, double, double) {}
template <class X> DEV void set_frame(Best<true> &b, const X &xf, d3 ot, d3, double u, double v) {
  b.set_tan(xf.world_dir(ot));
  b.set_uv(u, v);
}

// Common tail of the transformed intersectors: world point, world distance, raw normal.
template <bool SHADOW, bool UV, class X>
DEV void commit(Best<UV> &b, const X &xf, const RayL &ray, d3 op, d3 on, int prim, double &closest_t) {
  const d3 pw = xf.world_point(op);
  const double t = len(pw - ray.o) / ray.dlen;
  b.t = t;
  b.prim = prim;
  closest_t = t;
  if (!SHADOW) {
    b.set_p(pw);
    b.set_n(xf.world_normal(on));
  }
}

// ---- sphere (Sphere.cpp:6-80): unit sphere in object space ---------------------------------------
template <bool SHADOW, bool UV, class X>
DEV void isect_sphere(const X &xf, const RayL &ray, double t_min, double &closest_t, Best<UV> &b, int prim) {
  const d3 o = xf.obj_point(ray.o), d = xf.obj_dir(ray.d);
  const double a = len2(d), hb = dot(o, d), c = len2(o) - 1.0;
  const double disc = hb * hb - a * c;
  if (disc < 0) return;
  const double sq = sqrt(disc);
  double root = (-hb - sq) / a;
  if (root < t_min || root > closest_t) {
    root = (-hb + sq) / a;
    if (root < t_min || root > closest_t) return;
  }
  const d3 op = o + d * root;
  commit<SHADOW, UV>(b, xf, ray, op, op, prim, closest_t);
  if (UV && !SHADOW) {
    const double theta = acos(op.z), phi = atan2(op.y, op.x);
    d3 ot = mk(-op.y, op.x, 0.0);
    d3 ob = mk(-op.z * op.x, -op.z * op.y, op.x * op.x + op.y * op.y);
    if (len2(ot) < 1e-6) {
      ot = mk(1, 0, 0);
      ob = mk(0, 1, 0);
    }
    set_frame(b, xf, ot, ob, (phi + M_PI) / (2.0 * M_PI), 1.0 - (theta / M_PI));
  }
}

// ---- cube (Cube.cpp:10-145): the box [-1,1]^3 ----------------------------------------------------
template <bool SHADOW, bool UV, class X, class P>
DEV void isect_cube(const X &xf, P scale, const RayL &ray, double t_min, double &closest_t, Best<UV> &b,
                    int prim) {
  const d3 o = xf.obj_point(ray.o), d = xf.obj_dir(ray.d);
  double t_near = t_min, t_far = closest_t;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double oi = comp(o, i), di = comp(d, i);
    if (same(di, 0.0)) {
      if (oi < -1.0 || oi > 1.0) return;
    } else {
      double t1 = (-1.0 - oi) / di, t2 = (1.0 - oi) / di;
      if (t1 > t2) {
        const double s = t1;
        t1 = t2;
        t2 = s;
      }
      t_near = (t_near < t1) ? t1 : t_near;  // std::max(t_near, t1)
      t_far = (t2 < t_far) ? t2 : t_far;     // std::min(t_far, t2)
      if (t_near > t_far) return;
    }
  }
  if (t_near < t_min || t_near > closest_t) return;
  const d3 op = o + d * t_near;
  const double tol = 1e-4;
  d3 on;
  if (same(op.x, 1.0) || fabs(op.x - 1.0) < tol) on = mk(1, 0, 0);
  else if (same(op.x, -1.0) || fabs(op.x + 1.0) < tol) on = mk(-1, 0, 0);
  else if (same(op.y, 1.0) || fabs(op.y - 1.0) < tol) on = mk(0, 1, 0);
  else if (same(op.y, -1.0) || fabs(op.y + 1.0) < tol) on = mk(0, -1, 0);
  else if (same(op.z, 1.0) || fabs(op.z - 1.0) < tol) on = mk(0, 0, 1);
  else on = mk(0, 0, -1);
  commit<SHADOW, UV>(b, xf, ray, op, on, prim, closest_t);
  if (UV && !SHADOW) {
    double u, v;
    d3 ot, ob;
    if (fabs(on.x) > 0.5) {
      u = (op.z + 1.0) * scale[2];
      v = (op.y + 1.0) * scale[1];
      ot = mk(0, 0, on.x > 0 ? 1 : -1);
      ob = mk(0, 1, 0);
    } else if (fabs(on.y) > 0.5) {
      u = (op.x + 1.0) * scale[0];
      v = (op.z + 1.0) * scale[2];
      ot = mk(1, 0, 0);
      ob = mk(0, 0, on.y > 0 ? 1 : -1);
    } else {
      u = (op.x + 1.0) * scale[0];
      v = (op.y + 1.0) * scale[1];
      ot = mk(on.z > 0 ? 1 : -1, 0, 0);
      ob = mk(0, 1, 0);
    }
    set_frame(b, xf, ot, ob, u, v);
  }
}

// ---- plane (Plane.cpp:8-132): world-space rectangle, no transform, no motion ---------------------
DEV void plane_frame(Best<false> &, ro_doubles, d3) {}
DEV void plane_frame(Best<true> &b, ro_doubles pl, d3 ip) {
  const d3 e1 = ld3(pl + 12), e2 = ld3(pl + 16);
  const d3 local = ip - ld3(pl);
  b.set_uv(dot(local, e1) / pl[15], dot(local, e2) / pl[19]);
  b.set_tan(e1);
}
// pv: the first twelve values of the record (point, normal, bounds), wherever the caller holds them;
// pl: the record itself, read for the edge frame of a hit when texture coordinates are wanted
template <bool SHADOW, bool UV, class P>
DEV void isect_plane(P pv, ro_doubles pl, int flags, const RayL &ray, double t_min, double &closest_t, Best<UV> &b,
                     int prim) {
  const d3 p0 = mk(pv[0], pv[1], pv[2]), n = mk(pv[3], pv[4], pv[5]), lo = mk(pv[6], pv[7], pv[8]),
           hi = mk(pv[9], pv[10], pv[11]);
  const double denom = dot(n, ray.d);
  if (same(denom, 0.0)) return;
  const double t = dot(p0 - ray.o, n) / denom;
  if (t < t_min || t > closest_t) return;
  const d3 ip = ray.o + ray.d * t;
  const double tol = 1e-6;
  if ((flags & 1) && (ip.x < lo.x - tol || ip.x > hi.x + tol)) return;
  if ((flags & 2) && (ip.y < lo.y - tol || ip.y > hi.y + tol)) return;
  if ((flags & 4) && (ip.z < lo.z - tol || ip.z > hi.z + tol)) return;
  b.t = t;
  b.prim = prim;
  closest_t = t;
  if (!SHADOW) {
    b.set_p(ip);
    b.set_n(n);
    if (UV) plane_frame(b, pl, ip);
  }
}

// ---- torus (Torus.cpp:14-306) ---------------------------------------------------------------------
// Ferrari's method with the reference's clamps.  The reference evaluates the solver in x87
// long double; here it runs in double and the root it selects is then polished by the same
// double-precision Newton iteration on the implicit surface as in the reference (Torus.cpp:220-254),
// which is what fixes the returned distance.
#define TORUS_PEPS 1e-10
// cos on [0, 1.1] and acos on [-1, 1] for the three-real-root case of the resolvent cubic: the
// argument ranges are known, so neither needs the general routines' argument reduction (which
// costs the kernels some 36 vector registers at their peak).  Both are accurate to about 1 ulp.
DEV double cos_small(double x) {
  const double z = x * x;
  double p = 1.0 / 2432902008176640000.0;
  p = p * z - 1.0 / 6402373705728000.0;
  p = p * z + 1.0 / 20922789888000.0;
  p = p * z - 1.0 / 87178291200.0;
  p = p * z + 1.0 / 479001600.0;
  p = p * z - 1.0 / 3628800.0;
  p = p * z + 1.0 / 40320.0;
  p = p * z - 1.0 / 720.0;
  p = p * z + 1.0 / 24.0;
  p = p * z - 0.5;
  return p * z + 1.0;
}
DEV double acos_unit(double x) {
  const double ax = fabs(x);
  const bool small = ax < 0.5;
  const double z = small ? x * x : (1.0 - ax) * 0.5;
  const double pn = z * (1.66666666666666657415e-01 +
                         z * (-3.25565818622400915405e-01 +
                              z * (2.01212532134862925881e-01 +
                                   z * (-4.00555345006794114027e-02 +
                                        z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
  const double qn = 1.0 + z * (-2.40339491173441421878e+00 +
                               z * (2.02094576023350569471e+00 +
                                    z * (-6.88283971605453293030e-01 + z * 7.70381505559019352791e-02)));
  const double r = pn / qn;
  const double s = small ? x : sqrt(z);
  const double v = s + s * r;  // asin(x) (small) or asin(sqrt(z))
  if (small) return 1.57079632679489661923 - v;
  return x < 0 ? 3.14159265358979323846 - 2.0 * v : 2.0 * v;
}
// Double-double helpers (value = hi + lo, |lo| <= ulp(hi)/2) for the Taylor shift below.
struct dd {
  double hi, lo;
};
DEV dd dd_add_d(dd a, double b) {  // a + b
  const double s = a.hi + b, bb = s - a.hi;
  const double e = ((a.hi - (s - bb)) + (b - bb)) + a.lo;
  const double hi = s + e;
  return dd{hi, e - (hi - s)};
}
DEV dd dd_add(dd a, dd b) {
  const double s = a.hi + b.hi, bb = s - a.hi;
  const double e = ((a.hi - (s - bb)) + (b.hi - bb)) + (a.lo + b.lo);
  const double hi = s + e;
  return dd{hi, e - (hi - s)};
}
DEV dd dd_mul_d(dd a, double b) {  // a * b, the product of the leading parts exact through one fused multiply-add
  const double p = a.hi * b;
  const double e = __builtin_fma(a.hi, b, -p) + a.lo * b;
  const double hi = p + e;
  return dd{hi, e - (hi - p)};
}
// c[0..4] := coefficients of p(x + h), by repeated synthetic division in double-double.
DEV void taylor_shift_quartic(double *c, double h) {
  dd a[5];
#pragma unroll
  for (int i = 0; i < 5; i++) a[i] = dd{c[i], 0.0};
#pragma unroll
  for (int pass = 0; pass < 4; pass++) {
#pragma unroll
    for (int i = 3; i >= pass; i--) a[i] = dd_add(dd_mul_d(a[i + 1], h), a[i]);
  }
#pragma unroll
  for (int i = 0; i < 5; i++) c[i] = a[i].hi + a[i].lo;
}
DEV int quad_roots(double b, double c, double *r) {  // x^2 + b x + c  (Torus.cpp:18-34)
  double disc = b * b - 4.0 * c;
  if (disc < -TORUS_PEPS) return 0;
  if (disc < 0) disc = 0;
  if (disc == 0) {
    r[0] = -0.5 * b;
    return 1;
  }
  const double sq = sqrt(disc);
  const double q = -0.5 * (b + (b > 0 ? sq : -sq));
  r[0] = q;
  r[1] = c / q;
  return 2;
}
DEV double cubic_first_root(double a, double b, double c) {  // x^3 + a x^2 + b x + c  (Torus.cpp:38-73), roots[0]
  const double third = 1.0 / 3.0;
  const double sq_a = a * a;
  const double p = b - third * sq_a;
  const double q = a * (2.0 / 27.0 * sq_a - third * b) + c;
  const double cube_p = p * p * p;
  double D = q * q + 4.0 / 27.0 * cube_p;
  if (fabs(D) < TORUS_PEPS) D = 0;
  if (D > 0) {
    const double sD = sqrt(D);
    return cbrt(-0.5 * q + 0.5 * sD) + cbrt(-0.5 * q - 0.5 * sD) - third * a;
  }
  if (D == 0) return 2.0 * cbrt(-0.5 * q) - third * a;
  double arg = -0.5 * q / sqrt(-cube_p / 27.0);
  arg = arg < -1.0 ? -1.0 : (arg > 1.0 ? 1.0 : arg);
  return 2.0 * sqrt(-p / 3.0) * cos_small(acos_unit(arg) * third) - third * a;
}
DEV int quartic_roots(const double *c, double *s) {  // c[0] + ... + c[4] x^4  (Torus.cpp:77-145)
  if (fabs(c[4]) < TORUS_PEPS) return 0;
  const double inv_a = 1.0 / c[4];
  const double A = c[3] * inv_a, B = c[2] * inv_a, C = c[1] * inv_a, D = c[0] * inv_a;
  const double sq_A = A * A;
  const double p = -0.375 * sq_A + B;
  const double q = 0.125 * sq_A * A - 0.5 * A * B + C;
  const double r = -0.01171875 * sq_A * sq_A + 0.0625 * sq_A * B - 0.25 * A * C + D;
  int n = 0;
  double qr[2];
  if (fabs(q) < TORUS_PEPS) {
    const int k = quad_roots(p, r, qr);
    for (int i = 0; i < k; i++)
      if (qr[i] >= 0) {
        const double y = sqrt(qr[i]);
        s[n++] = y - 0.25 * A;
        s[n++] = -y - 0.25 * A;
      }
  } else {
    double z = cubic_first_root(2.0 * p, p * p - 4.0 * r, -q * q);
    if (z < 0 && z > -1e-5) z = 0;
    const double sz = sqrt(z > 0.0 ? z : 0.0);
    const double qz = fabs(sz) > TORUS_PEPS ? q / sz : 0.0;
    const double r1 = 0.5 * (p + z + qz), r2 = 0.5 * (p + z - qz);
    int k = quad_roots(-sz, r1, qr);
    for (int i = 0; i < k; i++) s[n++] = qr[i] - 0.25 * A;
    k = quad_roots(sz, r2, qr);
    for (int i = 0; i < k; i++) s[n++] = qr[i] - 0.25 * A;
  }
  return n;
}

template <bool SHADOW, bool UV, class X, class P>
DEV void isect_torus(const X &xf, P prm, const RayL &ray, double t_min, double &closest_t, Best<UV> &b,
                     int prim) {
  const d3 o = xf.obj_point(ray.o);
  d3 d = xf.obj_dir(ray.d);
  const double dl = len(d);
  d = d / dl;
  const double R = prm[0], rt = prm[1];
  const double R2 = R * R, r2 = rt * rt, fR2 = 4.0 * R2;
  const double tot = R + rt;
  const double oo = dot(o, o), od = dot(o, d);
  const double cs = oo - tot * tot;
  if (cs > 0 && od > 0) return;
  if (od * od - cs < 0) return;
  // The quartic's coefficients in the ray parameter measured from the ray origin, computed in double
  // exactly as the reference computes them (Torus.cpp:186-199).
  const double beta = 2 * od, gamma = oo - r2 - R2;
  double c[5], roots[4];
  c[4] = 1.0;
  c[3] = 2 * beta;
  c[2] = beta * beta + 2 * gamma + fR2 * d.z * d.z;
  c[1] = 2 * beta * gamma + 2 * fR2 * o.z * d.z;
  c[0] = gamma * gamma + fR2 * (o.z * o.z - r2);
  // The reference then finds the roots in x87 long double.  Seen from far away they sit in a tight
  // cluster hundreds of torus sizes out, which Ferrari's method in plain double cannot resolve; for
  // such rays the polynomial is first re-centred on the point of the ray nearest the torus centre,
  // with the Taylor shift carried out in double-double (exact to ~1e-32), and the re-centred, now
  // well-conditioned quartic is solved in double.  The roots are those of the reference's own
  // polynomial, rounding of its coefficients included.  Closer than a hundred torus sizes plain
  // double resolves the roots to better than 1e-7 of the torus size and the polynomial is solved as it is.
  const bool far = oo > 1.0e4 * (tot * tot);
  const double shift = far ? -od : 0.0;
  if (far) taylor_shift_quartic(c, shift);
  const int nr = quartic_roots(c, roots);
  const double lo = t_min * dl, hi = closest_t * dl;
  double cur = hi;
  bool any = false;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const double root = roots[i] + shift;
    if (i < nr && root >= lo && root < cur) {
      cur = root;
      any = true;
    }
  }
  if (!any) return;
  double tr = cur;
  for (int i = 0; i < 3; i++) {
    const d3 q = o + d * tr;
    const double xy = q.x * q.x + q.y * q.y;
    const double term = xy + q.z * q.z + R2 - r2;
    const double val = term * term - fR2 * xy;
    if (fabs(val) < 1e-10) break;
    const double cm = 4.0 * term;
    const double der = (cm * q.x - 2 * fR2 * q.x) * d.x + (cm * q.y - 2 * fR2 * q.y) * d.y + cm * q.z * d.z;
    if (fabs(der) < 1e-8) break;
    const double step = val / der;
    tr -= step;
    if (fabs(step) < 1e-6) break;
  }
  if (tr < lo || tr > hi) return;
  const d3 hp = o + d * tr;
  const d3 pw = xf.world_point(hp);
  const double t = len(pw - ray.o);  // the reference assumes a unit world direction here (Torus.cpp:284-285)
  b.t = t;
  b.prim = prim;
  closest_t = t;
  if (!SHADOW) {
    const double rho = sqrt(hp.x * hp.x + hp.y * hp.y);
    const double rs = rho > 1e-10 ? rho : 1e-10;
    const double rad = (rho - R) / rs;
    const d3 on = unit(mk(hp.x * rad, hp.y * rad, hp.z));
    b.set_p(pw);
    b.set_n(xf.world_normal(on));
    if (UV) {
      const double phi = atan2(hp.y, hp.x), theta = atan2(hp.z, rho - R);
      d3 ot = mk(-hp.y, hp.x, 0);
      if (len2(ot) < 1e-6) ot = mk(1, 0, 0);
      ot = unit(ot);
      const d3 ob = unit(cross(on, ot));
      set_frame(b, xf, ot, ob, (phi + M_PI) / (2.0 * M_PI), (theta + M_PI) / (2.0 * M_PI));
    }
  }
}

// ---- cylinder (Cylinder.cpp:7-116): side + both caps, strict `<` against the nearest so far --------
template <bool SHADOW, bool UV, class X, class P>
DEV void isect_cylinder(const X &xf, P prm, const RayL &ray, double t_min, double &closest_t,
                        Best<UV> &b, int prim) {
  const d3 o = xf.obj_point(ray.o), d = xf.obj_dir(ray.d);
  const double radius = prm[0], depth = prm[1], hd = depth / 2.0;
  const double a = d.x * d.x + d.y * d.y;
  const double bb = 2.0 * (o.x * d.x + o.y * d.y);
  const double c = o.x * o.x + o.y * o.y - radius * radius;
  double t_near = closest_t;
  int which = 0;  // 1 side, 2 top cap, 3 bottom cap
  if (fabs(a) > 1e-6) {
    const double disc = bb * bb - 4 * a * c;
    if (disc >= 0) {
      const double sq = sqrt(disc);
      const double t1 = (-bb - sq) / (2 * a), t2 = (-bb + sq) / (2 * a);
      if (t1 >= t_min && t1 < t_near) {
        const double z = o.z + t1 * d.z;
        if (z >= -hd && z <= hd) {
          t_near = t1;
          which = 1;
        }
      }
      if (t2 >= t_min && t2 < t_near) {
        const double z = o.z + t2 * d.z;
        if (z >= -hd && z <= hd) {
          t_near = t2;
          which = 1;
        }
      }
    }
  }
  if (fabs(d.z) > 1e-6) {
    const double tt = (hd - o.z) / d.z;
    if (tt >= t_min && tt < t_near) {
      const double x = o.x + tt * d.x, y = o.y + tt * d.y;
      if (x * x + y * y <= radius * radius) {
        t_near = tt;
        which = 2;
      }
    }
    const double tb = (-hd - o.z) / d.z;
    if (tb >= t_min && tb < t_near) {
      const double x = o.x + tb * d.x, y = o.y + tb * d.y;
      if (x * x + y * y <= radius * radius) {
        t_near = tb;
        which = 3;
      }
    }
  }
  if (!which) return;
  const d3 hp = o + d * t_near;
  const d3 on = which == 1 ? mk(hp.x / radius, hp.y / radius, 0.0) : mk(0, 0, which == 2 ? 1 : -1);
  commit<SHADOW, UV>(b, xf, ray, hp, on, prim, closest_t);
  if (UV && !SHADOW) {
    double u, v;
    d3 ot, ob;
    if (which == 1) {
      const double z = o.z + t_near * d.z;
      u = (atan2(hp.y, hp.x) + M_PI) / (2.0 * M_PI);
      v = (z + hd) / depth;
      ot = unit(mk(-hp.y, hp.x, 0));
      ob = mk(0, 0, 1);
    } else {
      const double x = o.x + t_near * d.x, y = o.y + t_near * d.y;
      u = (x / radius + 1.0) * 0.5;
      v = (y / radius + 1.0) * 0.5;
      ot = mk(1, 0, 0);
      ob = mk(0, 1, 0);
    }
    set_frame(b, xf, ot, ob, u, v);
  }
}

// ---- cone (Cone.cpp:13-171): apex at +depth/2, base cap at -depth/2 -------------------------------
template <bool SHADOW, bool UV, class X, class P>
DEV void isect_cone(const X &xf, P prm, const RayL &ray, double t_min, double &closest_t, Best<UV> &b,
                    int prim) {
  const double EPS = 1e-6;
  const d3 o = xf.obj_point(ray.o), d = xf.obj_dir(ray.d);
  const double radius = prm[0], height = prm[1], hd = height * 0.5;
  const double a_cyl = d.x * d.x + d.y * d.y;
  const double b_cyl = o.x * d.x + o.y * d.y;
  const double c_cyl = o.x * o.x + o.y * o.y - radius * radius;
  if (c_cyl > 0 && b_cyl > 0 && a_cyl > EPS) return;
  if (a_cyl > EPS && b_cyl * b_cyl - a_cyl * c_cyl < 0) return;
  const double t_max = closest_t;
  if (fabs(d.z) > EPS) {
    const double inv_dz = 1.0 / d.z;
    double z0 = (-hd - o.z) * inv_dz, z1 = (hd - o.z) * inv_dz;
    if (z0 > z1) {
      const double s = z0;
      z0 = z1;
      z1 = s;
    }
    if (z1 < t_min || z0 > t_max) return;
  } else if (o.z < -hd || o.z > hd) {
    return;
  }
  const double k = radius / height, k2 = k * k;
  const double zt = hd - o.z;
  const double a = a_cyl - k2 * d.z * d.z;
  const double bq = 2.0 * (b_cyl + k2 * zt * d.z);
  const double c = (o.x * o.x + o.y * o.y) - k2 * zt * zt;
  double t_near = t_max;
  int which = 0;  // 1 side, 2 cap
  if (fabs(a) > EPS) {
    const double disc = bq * bq - 4.0 * a * c;
    if (disc >= 0) {
      const double sq = sqrt(disc), i2a = 0.5 / a;
      double t1 = (-bq - sq) * i2a, t2 = (-bq + sq) * i2a;
      if (t1 > t2) {
        const double s = t1;
        t1 = t2;
        t2 = s;
      }
      if (t1 >= t_min && t1 <= t_max) {
        const double z = o.z + t1 * d.z;
        if (z >= -hd && z <= hd) {
          t_near = t1;
          which = 1;
        }
      }
      if (!which && t2 >= t_min && t2 < t_near) {
        const double z = o.z + t2 * d.z;
        if (z >= -hd && z <= hd) {
          t_near = t2;
          which = 1;
        }
      }
    }
  }
  double cx = 0, cy = 0;
  if (fabs(d.z) > EPS) {
    const double tc = (-hd - o.z) * (1.0 / d.z);
    if (tc >= t_min && tc < t_near) {
      cx = o.x + tc * d.x;
      cy = o.y + tc * d.y;
      if (cx * cx + cy * cy <= radius * radius) {
        t_near = tc;
        which = 2;
      }
    }
  }
  if (!which) return;
  d3 hp, on;
  if (which == 1) {
    hp = o + d * t_near;
    on = unit(mk(hp.x, hp.y, k2 * (hd - hp.z)));
  } else {
    hp = mk(cx, cy, -hd);
    on = mk(0, 0, -1);
  }
  const d3 pw = xf.world_point(hp);
  const double t = sqrt(len2(pw - ray.o) / len2(ray.d));  // Cone.cpp:138-140
  b.t = t;
  b.prim = prim;
  closest_t = t;
  if (!SHADOW) {
    b.set_p(pw);
    b.set_n(xf.world_normal(on));
    if (UV) {
      double u, v;
      d3 ot, ob;
      if (which == 1) {
        const double z = o.z + t_near * d.z;
        u = (atan2(hp.y, hp.x) + M_PI) * (0.5 / M_PI);
        v = (z + hd) / height;
      } else {
        const double ir = 1.0 / radius;
        u = (cx * ir + 1.0) * 0.5;
        v = (cy * ir + 1.0) * 0.5;
      }
      if (fabs(on.z) > 0.9) {
        ot = mk(1, 0, 0);
        ob = mk(0, 1, 0);
      } else {
        ot = mk(-hp.y, hp.x, 0);
        const double l2 = len2(ot);
        ot = l2 > EPS ? ot * (1.0 / sqrt(l2)) : mk(1, 0, 0);
        ob = cross(on, ot);
      }
      set_frame(b, xf, ot, ob, u, v);
    }
  }
}

// One primitive against one ray, with the transform kind chosen by the (wave-uniform) record.
template <bool SHADOW, bool UV, bool MOTION, class X, class P>
DEV void isect_shape(int type, P param, const X &xf, const RayL &ray, double t_min, double &closest_t, Best<UV> &b,
                     int prim) {
  switch (type) {
    case MIRT_SPHERE: isect_sphere<SHADOW, UV>(xf, ray, t_min, closest_t, b, prim); break;
    case MIRT_CUBE: isect_cube<SHADOW, UV>(xf, param, ray, t_min, closest_t, b, prim); break;
    case MIRT_TORUS: isect_torus<SHADOW, UV>(xf, param, ray, t_min, closest_t, b, prim); break;
    case MIRT_CYLINDER: isect_cylinder<SHADOW, UV>(xf, param, ray, t_min, closest_t, b, prim); break;
    case MIRT_CONE: isect_cone<SHADOW, UV>(xf, param, ray, t_min, closest_t, b, prim); break;
    default: break;
  }
}

// The same for the per-lane walks, where every lane holds a different primitive and a record
// field read where it is first needed is a round trip to memory of its own (flags, then kind, then
// the map, then the parameters ...).  Here what every test needs -- the 32-byte header, the
// world-to-object rows (planes: point, normal, bounds) and the parameters, 160 bytes -- is fetched
// by ten loads issued together before anything is looked at; the object-to-world rows are read
// only by a ray that hits.
struct XfSV {
  double m[12];  // world_to_object rows 0-2
  ro_doubles w;  // object_to_world rows 0-2 (DPrim::xf + 12)
  DEV d3 obj_point(d3 p) const {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
  }
  DEV d3 obj_dir(d3 d) const {
    return {m[0] * d.x + m[1] * d.y + m[2] * d.z, m[4] * d.x + m[5] * d.y + m[6] * d.z,
            m[8] * d.x + m[9] * d.y + m[10] * d.z};
  }
  DEV d3 world_point(d3 p) const {
    return {w[0] * p.x + w[1] * p.y + w[2] * p.z + w[3], w[4] * p.x + w[5] * p.y + w[6] * p.z + w[7],
            w[8] * p.x + w[9] * p.y + w[10] * p.z + w[11]};
  }
  DEV d3 world_dir(d3 d) const {
    return {w[0] * d.x + w[1] * d.y + w[2] * d.z, w[4] * d.x + w[5] * d.y + w[6] * d.z,
            w[8] * d.x + w[9] * d.y + w[10] * d.z};
  }
  DEV d3 world_normal(d3 n) const {
    return {m[0] * n.x + m[4] * n.y + m[8] * n.z, m[1] * n.x + m[5] * n.y + m[9] * n.z,
            m[2] * n.x + m[6] * n.y + m[10] * n.z};
  }
};
typedef double gdbl2 __attribute__((ext_vector_type(2)));
typedef double gdbl4 __attribute__((ext_vector_type(4)));
typedef int gint4 __attribute__((ext_vector_type(4)));
typedef int gint8 __attribute__((ext_vector_type(8)));

// intersect_obj (Bvh.cpp:244-304) for one ray, on a record already fetched: header words (type,
// visible, motion, material, motion_index, plane_flags), the first twelve values of DPrim::xf, the
// parameters; `rec` is the record itself for what only a hit (or a moving primitive) reads.
// MOTION = false compiles the kernels used when every ray of the launch is at shutter time 0.5
// (shadow rays, motion blur off).
template <bool SHADOW, bool UV, bool MOTION>
DEV void isect_fetched(const DScene &sc, int type, int visible, int motion, int motion_index, int plane_flags,
                       const double *xv, const double *param, const RO DPrim *rec, int prim, const RayL &ray,
                       double t_min, double &closest_t, Best<UV> &b) {
  if (!visible) return;
  if (type == MIRT_PLANE) {
    isect_plane<SHADOW, UV>(xv, rec->xf, plane_flags, ray, t_min, closest_t, b, prim);
    return;
  }
  if (MOTION && motion != MOTION_NONE) {
    const RO DMotion &mo = rdonly(sc.motion)[motion_index];
    if (motion == MOTION_TRANSLATE) {
      XfT xf;
      xf.m = mo.m;
      const double t = ray.time;
      xf.t = mk(mo.t0[0] + mo.dt[0] * t, mo.t0[1] + mo.dt[1] * t, mo.t0[2] + mo.dt[2] * t);
      isect_shape<SHADOW, UV, MOTION>(type, param, xf, ray, t_min, closest_t, b, prim);
    } else {
      const XfG xf = motion_general(mo, ray.time);
      isect_shape<SHADOW, UV, MOTION>(type, param, xf, ray, t_min, closest_t, b, prim);
    }
    return;
  }
  XfSV xf;
#pragma unroll
  for (int i = 0; i < 12; i++) xf.m[i] = xv[i];
  xf.w = rec->xf + 12;
  isect_shape<SHADOW, UV, MOTION>(type, param, xf, ray, t_min, closest_t, b, prim);
}

// Per-lane walks: every lane holds a different primitive; ten vector loads issued together.
template <bool SHADOW, bool UV, bool MOTION>
DEV void isect_prim_lane(const DScene &sc, const DPrim *pr, int prim, const RayL &ray, double t_min,
                         double &closest_t, Best<UV> &b) {
  const gint4 *hq = reinterpret_cast<const gint4 *>(pr);
  gint4 h0 = hq[0], h1 = hq[1];  // type, visible, motion, material | motion_index, plane_flags, -, -
  const gdbl2 *xq = reinterpret_cast<const gdbl2 *>(pr->xf);
  gdbl2 x0 = xq[0], x1 = xq[1], x2 = xq[2], x3 = xq[3], x4 = xq[4], x5 = xq[5];
  const gdbl2 *pq = reinterpret_cast<const gdbl2 *>(pr->param);
  gdbl2 p0 = pq[0], p1 = pq[1];
  // everything has to have arrived here: keeps the loads above from being moved behind the branches below
  asm volatile("" : "+v"(h0), "+v"(h1), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(p0), "+v"(p1));
  const double xv[12] = {x0.x, x0.y, x1.x, x1.y, x2.x, x2.y, x3.x, x3.y, x4.x, x4.y, x5.x, x5.y};
  const double param[4] = {p0.x, p0.y, p1.x, p1.y};
  isect_fetched<SHADOW, UV, MOTION>(sc, h0.x, h0.y, h0.z, h1.x, h1.y, xv, param, rdonly(pr), prim, ray, t_min,
                                    closest_t, b);
}

// Packet walk: the record's address is wave-uniform, so the same 160 bytes come through the scalar
// data path -- five loads issued together, values in scalar registers.
template <bool SHADOW, bool UV, bool MOTION>
DEV void isect_prim(const DScene &sc, const RO DPrim *pr, int prim, const RayL &ray, double t_min, double &closest_t,
                    Best<UV> &b) {
  gint8 h = *reinterpret_cast<const RO gint8 *>(pr);
  const RO gdbl4 *xq = reinterpret_cast<const RO gdbl4 *>(pr->xf);
  gdbl4 x0 = xq[0], x1 = xq[1], x2 = xq[2];
  gdbl4 p0 = *reinterpret_cast<const RO gdbl4 *>(pr->param);
  asm volatile("" : "+s"(h), "+s"(x0), "+s"(x1), "+s"(x2), "+s"(p0));
  const double xv[12] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
  const double param[4] = {p0.x, p0.y, p0.z, p0.w};
  isect_fetched<SHADOW, UV, MOTION>(sc, h.s0, h.s1, h.s2, h.s4, h.s5, xv, param, pr, prim, ray, t_min, closest_t, b);
}
