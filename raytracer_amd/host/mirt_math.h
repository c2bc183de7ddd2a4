// Host-side linear algebra of the drop-in surface: the names and semantics host code written
// against the reference renderer expects (Vec3/Point/Direction/Color of Code/Math/Vector.h:111-212,
// Mat4 and Transform of Code/Math/Transform.h:12-181, Quaternion of Code/Math/Quaternion.h:9-136).
// Only what the host needs to parse a scene, build the BVH and flatten it lives here; all per-ray
// arithmetic runs on the GPU.
#pragma once

#include <cmath>
#include <cstddef>
#include <ostream>

struct Vec3 {
  double x = 0.0, y = 0.0, z = 0.0;

  Vec3() = default;
  Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}

  double operator[](std::size_t i) const { return i == 0 ? x : (i == 1 ? y : z); }
  double &operator[](std::size_t i) { return i == 0 ? x : (i == 1 ? y : z); }

  double r() const { return x; }
  double g() const { return y; }
  double b() const { return z; }
  double &r() { return x; }
  double &g() { return y; }
  double &b() { return z; }

  Vec3 operator+(const Vec3 &o) const { return {x + o.x, y + o.y, z + o.z}; }
  Vec3 operator-(const Vec3 &o) const { return {x - o.x, y - o.y, z - o.z}; }
  Vec3 operator*(const Vec3 &o) const { return {x * o.x, y * o.y, z * o.z}; }
  Vec3 operator*(double c) const { return {x * c, y * c, z * c}; }
  Vec3 operator/(double c) const { return {x / c, y / c, z / c}; }
  Vec3 operator-() const { return {-x, -y, -z}; }
  Vec3 &operator+=(const Vec3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
  Vec3 &operator-=(const Vec3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
  Vec3 &operator*=(double c) { x *= c; y *= c; z *= c; return *this; }
  Vec3 &operator/=(double c) { x /= c; y /= c; z /= c; return *this; }
  friend Vec3 operator*(double c, const Vec3 &v) { return v * c; }

  double dot(const Vec3 &o) const { return x * o.x + y * o.y + z * o.z; }
  Vec3 cross(const Vec3 &o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
  double length_squared() const { return x * x + y * y + z * z; }
  double length() const { return std::sqrt(length_squared()); }
  Vec3 norm() const { return *this / length(); }

  friend std::ostream &operator<<(std::ostream &os, const Vec3 &v) {
    return os << "(" << v.x << ", " << v.y << ", " << v.z << ")";
  }
};

using Point = Vec3;
using Direction = Vec3;
using Color = Vec3;
using Pixel = Vec3;

struct Quaternion {
  double w = 1.0, x = 0.0, y = 0.0, z = 0.0;
  Quaternion() = default;
  Quaternion(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
  double dot(const Quaternion &q) const { return w * q.w + x * q.x + y * q.y + z * q.z; }
  double length() const { return std::sqrt(w * w + x * x + y * y + z * z); }
  Quaternion norm() const {
    const double len = length();
    return len < 1e-10 ? Quaternion() : Quaternion(w / len, x / len, y / len, z / len);
  }
  static Quaternion slerp(const Quaternion &a, const Quaternion &b, double t);
  void to_matrix(double rot[3][3]) const;
};

struct BoundingBox {
  Point min, max;
  void grow(const BoundingBox &o);
};

// Row-major homogeneous 4x4; identity by default.
struct Mat4 {
  double m[4][4];
  Mat4();
  Mat4 operator*(const Mat4 &o) const;
  Point transform_point(const Point &p) const;
  Direction transform_direction(const Direction &d) const;
  Direction transform_normal(const Direction &n) const;
  Mat4 inverse() const;

  static Mat4 translate(const Vec3 &t);
  static Mat4 scale(double sx, double sy, double sz);
  static Mat4 scale_uniform(double s) { return scale(s, s, s); }
  static Mat4 rotate_x(double a);
  static Mat4 rotate_y(double a);
  static Mat4 rotate_z(double a);
  static Mat4 rotate_euler(const Vec3 &e) { return rotate_z(e.z) * rotate_y(e.y) * rotate_x(e.x); }

  Vec3 extract_translation() const { return {m[0][3], m[1][3], m[2][3]}; }
  Vec3 extract_scale() const;
  Quaternion extract_rotation() const;
  static Mat4 compose(const Vec3 &t, const Quaternion &r, const Vec3 &s);
  static Mat4 interpolate(const Mat4 &a, const Mat4 &b, double t);
};

struct Transform {
  Mat4 object_to_world;
  Mat4 world_to_object;
  Transform() = default;
  explicit Transform(const Mat4 &o2w) : object_to_world(o2w), world_to_object(o2w.inverse()) {}
  static Transform from_trs(const Point &t, const Point &euler, double s);
  static Transform from_trs_nonuniform(const Point &t, const Point &euler, const Vec3 &s);
  Point object_to_world_point(const Point &p) const { return object_to_world.transform_point(p); }
  Direction object_to_world_normal(const Direction &n) const { return world_to_object.transform_normal(n).norm(); }
  Direction object_to_world_direction(const Direction &d) const { return object_to_world.transform_direction(d).norm(); }
  BoundingBox transform_bbox(const BoundingBox &b) const;
};
