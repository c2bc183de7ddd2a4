// Host linear algebra (see mirt_math.h).  The arithmetic order of Mat4::operator*, Mat4::inverse
// and the TRS helpers follows the reference (Code/Math/Transform.h:25-36, Transform.cpp:8-64,
// 145-256, Quaternion.h:73-135) operation for operation, because the matrices computed here
// (cached_transform, bounding boxes) decide the BVH topology and are compared bit for bit with the
// reference's in the parity tests.
#include "mirt_math.h"

#include <algorithm>
#include <utility>

Mat4::Mat4() {
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) m[r][c] = (r == c) ? 1.0 : 0.0;
}

Mat4 Mat4::operator*(const Mat4 &o) const {
  Mat4 out;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double acc = 0;
      for (int k = 0; k < 4; ++k) acc += m[r][k] * o.m[k][c];
      out.m[r][c] = acc;
    }
  return out;
}

Point Mat4::transform_point(const Point &p) const {
  const double X = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3];
  const double Y = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3];
  const double Z = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3];
  const double W = m[3][0] * p.x + m[3][1] * p.y + m[3][2] * p.z + m[3][3];
  if (W != 1.0 && W != 0.0) return {X / W, Y / W, Z / W};
  return {X, Y, Z};
}

Direction Mat4::transform_direction(const Direction &d) const {
  return {m[0][0] * d.x + m[0][1] * d.y + m[0][2] * d.z, m[1][0] * d.x + m[1][1] * d.y + m[1][2] * d.z,
          m[2][0] * d.x + m[2][1] * d.y + m[2][2] * d.z};
}

// Multiplies by the transpose (a normal is a covector); called on world_to_object.
Direction Mat4::transform_normal(const Direction &n) const {
  return Direction(m[0][0] * n.x + m[1][0] * n.y + m[2][0] * n.z, m[0][1] * n.x + m[1][1] * n.y + m[2][1] * n.z,
                   m[0][2] * n.x + m[1][2] * n.y + m[2][2] * n.z)
      .norm();
}

// Gauss-Jordan on [A | I] with partial pivoting; a pivot below 1e-10 leaves its column alone.
Mat4 Mat4::inverse() const {
  double a[4][8];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      a[r][c] = m[r][c];
      a[r][c + 4] = (r == c) ? 1.0 : 0.0;
    }
  for (int col = 0; col < 4; ++col) {
    int best = col;
    for (int r = col + 1; r < 4; ++r)
      if (std::abs(a[r][col]) > std::abs(a[best][col])) best = r;
    if (best != col)
      for (int c = 0; c < 8; ++c) std::swap(a[col][c], a[best][c]);
    const double piv = a[col][col];
    if (std::abs(piv) < 1e-10) continue;
    for (int c = 0; c < 8; ++c) a[col][c] /= piv;
    for (int r = 0; r < 4; ++r) {
      if (r == col) continue;
      const double f = a[r][col];
      for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
    }
  }
  Mat4 out;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) out.m[r][c] = a[r][c + 4];
  return out;
}

Mat4 Mat4::translate(const Vec3 &t) {
  Mat4 o;
  o.m[0][3] = t.x;
  o.m[1][3] = t.y;
  o.m[2][3] = t.z;
  return o;
}

Mat4 Mat4::scale(double sx, double sy, double sz) {
  Mat4 o;
  o.m[0][0] = sx;
  o.m[1][1] = sy;
  o.m[2][2] = sz;
  return o;
}

Mat4 Mat4::rotate_x(double a) {
  Mat4 o;
  const double c = std::cos(a), s = std::sin(a);
  o.m[1][1] = c; o.m[1][2] = -s;
  o.m[2][1] = s; o.m[2][2] = c;
  return o;
}

Mat4 Mat4::rotate_y(double a) {
  Mat4 o;
  const double c = std::cos(a), s = std::sin(a);
  o.m[0][0] = c; o.m[0][2] = s;
  o.m[2][0] = -s; o.m[2][2] = c;
  return o;
}

Mat4 Mat4::rotate_z(double a) {
  Mat4 o;
  const double c = std::cos(a), s = std::sin(a);
  o.m[0][0] = c; o.m[0][1] = -s;
  o.m[1][0] = s; o.m[1][1] = c;
  return o;
}

Vec3 Mat4::extract_scale() const {
  return {Vec3(m[0][0], m[1][0], m[2][0]).length(), Vec3(m[0][1], m[1][1], m[2][1]).length(),
          Vec3(m[0][2], m[1][2], m[2][2]).length()};
}

// Shepperd's method on the scale-free rotation block.
Quaternion Mat4::extract_rotation() const {
  const Vec3 s = extract_scale();
  if (s.x < 1e-10 || s.y < 1e-10 || s.z < 1e-10) return Quaternion();
  double R[3][3];
  for (int r = 0; r < 3; ++r) {
    R[r][0] = m[r][0] / s.x;
    R[r][1] = m[r][1] / s.y;
    R[r][2] = m[r][2] / s.z;
  }
  const double tr = R[0][0] + R[1][1] + R[2][2];
  if (tr > 0.0) {
    const double k = std::sqrt(tr + 1.0) * 2.0;
    return {0.25 * k, (R[2][1] - R[1][2]) / k, (R[0][2] - R[2][0]) / k, (R[1][0] - R[0][1]) / k};
  }
  if (R[0][0] > R[1][1] && R[0][0] > R[2][2]) {
    const double k = std::sqrt(1.0 + R[0][0] - R[1][1] - R[2][2]) * 2.0;
    return {(R[2][1] - R[1][2]) / k, 0.25 * k, (R[0][1] + R[1][0]) / k, (R[0][2] + R[2][0]) / k};
  }
  if (R[1][1] > R[2][2]) {
    const double k = std::sqrt(1.0 + R[1][1] - R[0][0] - R[2][2]) * 2.0;
    return {(R[0][2] - R[2][0]) / k, (R[0][1] + R[1][0]) / k, 0.25 * k, (R[1][2] + R[2][1]) / k};
  }
  const double k = std::sqrt(1.0 + R[2][2] - R[0][0] - R[1][1]) * 2.0;
  return {(R[1][0] - R[0][1]) / k, (R[0][2] + R[2][0]) / k, (R[1][2] + R[2][1]) / k, 0.25 * k};
}

void Quaternion::to_matrix(double rot[3][3]) const {
  const double xx = x * x, yy = y * y, zz = z * z;
  const double xy = x * y, xz = x * z, yz = y * z;
  const double wx = w * x, wy = w * y, wz = w * z;
  rot[0][0] = 1.0 - 2.0 * (yy + zz); rot[0][1] = 2.0 * (xy - wz);       rot[0][2] = 2.0 * (xz + wy);
  rot[1][0] = 2.0 * (xy + wz);       rot[1][1] = 1.0 - 2.0 * (xx + zz); rot[1][2] = 2.0 * (yz - wx);
  rot[2][0] = 2.0 * (xz - wy);       rot[2][1] = 2.0 * (yz + wx);       rot[2][2] = 1.0 - 2.0 * (xx + yy);
}

Quaternion Quaternion::slerp(const Quaternion &a, const Quaternion &b, double t) {
  const Quaternion p = a.norm();
  Quaternion q = b.norm();
  double d = p.dot(q);
  if (d < 0.0) {
    q = Quaternion(-q.w, -q.x, -q.y, -q.z);
    d = -d;
  }
  if (d > 0.9995)
    return Quaternion(p.w + (q.w - p.w) * t, p.x + (q.x - p.x) * t, p.y + (q.y - p.y) * t, p.z + (q.z - p.z) * t)
        .norm();
  d = std::min(1.0, std::max(-1.0, d));
  const double th0 = std::acos(d), th = th0 * t;
  const double sn = std::sin(th), sn0 = std::sin(th0);
  const double k0 = std::cos(th) - d * sn / sn0, k1 = sn / sn0;
  return {k0 * p.w + k1 * q.w, k0 * p.x + k1 * q.x, k0 * p.y + k1 * q.y, k0 * p.z + k1 * q.z};
}

Mat4 Mat4::compose(const Vec3 &t, const Quaternion &r, const Vec3 &s) {
  Mat4 o;
  double R[3][3];
  r.to_matrix(R);
  for (int i = 0; i < 3; ++i) {
    o.m[i][0] = R[i][0] * s.x;
    o.m[i][1] = R[i][1] * s.y;
    o.m[i][2] = R[i][2] * s.z;
  }
  o.m[0][3] = t.x;
  o.m[1][3] = t.y;
  o.m[2][3] = t.z;
  return o;
}

Mat4 Mat4::interpolate(const Mat4 &a, const Mat4 &b, double t) {
  const Vec3 ta = a.extract_translation(), sa = a.extract_scale();
  const Vec3 tb = b.extract_translation(), sb = b.extract_scale();
  return compose(ta + (tb - ta) * t, Quaternion::slerp(a.extract_rotation(), b.extract_rotation(), t),
                 sa + (sb - sa) * t);
}

Transform Transform::from_trs(const Point &t, const Point &euler, double s) {
  return Transform(Mat4::translate(t) * Mat4::rotate_euler(euler) * Mat4::scale_uniform(s));
}

Transform Transform::from_trs_nonuniform(const Point &t, const Point &euler, const Vec3 &s) {
  return Transform(Mat4::translate(t) * Mat4::rotate_euler(euler) * Mat4::scale(s.x, s.y, s.z));
}

void BoundingBox::grow(const BoundingBox &o) {
  min.x = std::min(min.x, o.min.x); min.y = std::min(min.y, o.min.y); min.z = std::min(min.z, o.min.z);
  max.x = std::max(max.x, o.max.x); max.y = std::max(max.y, o.max.y); max.z = std::max(max.z, o.max.z);
}

// World AABB of the eight transformed corners.
BoundingBox Transform::transform_bbox(const BoundingBox &b) const {
  BoundingBox out;
  for (int i = 0; i < 8; ++i) {
    const Point c((i & 1) ? b.max.x : b.min.x, (i & 2) ? b.max.y : b.min.y, (i & 4) ? b.max.z : b.min.z);
    const Point w = object_to_world.transform_point(c);
    if (i == 0) {
      out.min = out.max = w;
    } else {
      out.grow(BoundingBox{w, w});
    }
  }
  return out;
}
