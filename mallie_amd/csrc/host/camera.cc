// host/camera.cc -- mallie::Camera of the facade: the per-pass camera frame (host code, a few hundred flops).
//
// Follows Camera::BuildCameraFrame / GenerateRay (camera.cc:40-240) together with the helpers it calls:
// Matrix::LookAt / Inverse / Mult / MultV (matrix.cc:42-216) and build_rotmatrix (trackball.cc:268-291).
// The float intrusions are deliberate and load-bearing for bit parity: tanf() for the focal length and
// a float-rounded length in the frame's three normalisations.
#include <cmath>
#include <cstring>

#include "../../../include/mallie/mallie_api.hpp"
#include "../../../include/mgpu.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

typedef double Mat4[4][4];

inline double guarded_length(const double v[3]) { // 0 when the squared length is below 1e-30
  const double s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  return (std::fabs(s) > 1.0e-30) ? std::sqrt(s) : 0.0;
}

inline void cross(double out[3], const double a[3], const double b[3]) {
  out[0] = a[1] * b[2] - a[2] * b[1];
  out[1] = a[2] * b[0] - a[0] * b[2];
  out[2] = a[0] * b[1] - a[1] * b[0];
}

template <typename LenT> inline void unitize(double v[3]) { // LenT = double in matrix.cc, float in camera.cc
  const LenT len = (LenT)guarded_length(v);
  if (std::fabs(len) > 1.0e-30) {
    const double inv = 1.0 / len;
    v[0] *= inv;
    v[1] *= inv;
    v[2] *= inv;
  }
}

// rows 0..2 = right, up', -forward; row 3 = eye (the transposed convention of matrix.cc:79-97)
void look_at(Mat4 m, const double eye[3], const double target[3], const double up[3]) {
  double fwd[3] = {target[0] - eye[0], target[1] - eye[1], target[2] - eye[2]};
  double right[3], up2[3];
  unitize<double>(fwd);
  cross(right, fwd, up);
  unitize<double>(right);
  cross(up2, right, fwd);
  unitize<double>(up2);
  for (int c = 0; c < 3; c++) {
    m[0][c] = right[c];
    m[1][c] = up2[c];
    m[2][c] = -fwd[c];
    m[3][c] = eye[c];
  }
  m[0][3] = m[1][3] = m[2][3] = 0.0;
  m[3][3] = 1.0;
}

// In-place inverse by cofactors of the transposed matrix ("Cramer's rule" routine of matrix.cc:102-196).  A cofactor is
// (p_a*s_a + p_b*s_b + p_c*s_c) - (p_d*s_d + p_e*s_e + p_f*s_f) with p = products of two source entries; the term
// tables below list (pair, source) indices in the order the reference adds them, which fixes the rounding.
struct Term { unsigned char pair, src; };
struct Cofactor { Term add[3], sub[3]; };

const unsigned char kPairsHi[12][2] = {{10, 15}, {11, 14}, {9, 15}, {11, 13}, {9, 14}, {10, 13},
                                       {8, 15},  {11, 12}, {8, 14}, {10, 12}, {8, 13}, {9, 12}};
const unsigned char kPairsLo[12][2] = {{2, 7}, {3, 6}, {1, 7}, {3, 5}, {1, 6}, {2, 5},
                                       {0, 7}, {3, 4}, {0, 6}, {2, 4}, {0, 5}, {1, 4}};
const Cofactor kRows01[8] = {
    {{{0, 5}, {3, 6}, {4, 7}}, {{1, 5}, {2, 6}, {5, 7}}},   {{{1, 4}, {6, 6}, {9, 7}}, {{0, 4}, {7, 6}, {8, 7}}},
    {{{2, 4}, {7, 5}, {10, 7}}, {{3, 4}, {6, 5}, {11, 7}}}, {{{5, 4}, {8, 5}, {11, 6}}, {{4, 4}, {9, 5}, {10, 6}}},
    {{{1, 1}, {2, 2}, {5, 3}}, {{0, 1}, {3, 2}, {4, 3}}},   {{{0, 0}, {7, 2}, {8, 3}}, {{1, 0}, {6, 2}, {9, 3}}},
    {{{3, 0}, {6, 1}, {11, 3}}, {{2, 0}, {7, 1}, {10, 3}}}, {{{4, 0}, {9, 1}, {10, 2}}, {{5, 0}, {8, 1}, {11, 2}}}};
const Cofactor kRows23[8] = {
    {{{0, 13}, {3, 14}, {4, 15}}, {{1, 13}, {2, 14}, {5, 15}}},   {{{1, 12}, {6, 14}, {9, 15}}, {{0, 12}, {7, 14}, {8, 15}}},
    {{{2, 12}, {7, 13}, {10, 15}}, {{3, 12}, {6, 13}, {11, 15}}}, {{{5, 12}, {8, 13}, {11, 14}}, {{4, 12}, {9, 13}, {10, 14}}},
    {{{2, 10}, {5, 11}, {1, 9}}, {{4, 11}, {0, 9}, {3, 10}}},     {{{8, 11}, {0, 8}, {7, 10}}, {{6, 10}, {9, 11}, {1, 8}}},
    {{{6, 9}, {11, 11}, {3, 8}}, {{10, 11}, {2, 8}, {7, 9}}},     {{{10, 10}, {4, 8}, {9, 9}}, {{8, 9}, {11, 0}, {5, 8}}}};

void invert(Mat4 m) {
  double s[16], p[12];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) s[r + 4 * c] = m[r][c];
  for (int half = 0; half < 2; half++) {
    const unsigned char(*pairs)[2] = half ? kPairsLo : kPairsHi;
    const Cofactor *cof = half ? kRows23 : kRows01;
    for (int i = 0; i < 12; i++) p[i] = s[pairs[i][0]] * s[pairs[i][1]];
    for (int e = 0; e < 8; e++) {
      const Cofactor &c = cof[e];
      double plus = p[c.add[0].pair] * s[c.add[0].src] + p[c.add[1].pair] * s[c.add[1].src] + p[c.add[2].pair] * s[c.add[2].src];
      const double minus =
          p[c.sub[0].pair] * s[c.sub[0].src] + p[c.sub[1].pair] * s[c.sub[1].src] + p[c.sub[2].pair] * s[c.sub[2].src];
      plus -= minus;
      m[2 * half + e / 4][e % 4] = plus;
    }
  }
  double det = s[0] * m[0][0] + s[1] * m[0][1] + s[2] * m[0][2] + s[3] * m[0][3];
  det = 1.0 / det;
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) m[r][c] *= det;
}

// dst[i][j] = sum_k a[k][j] * b[i][k], summed onto 0 in k order (Matrix::Mult, matrix.cc:198-207)
void multiply(Mat4 dst, Mat4 a, Mat4 b) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double sum = 0;
      for (int k = 0; k < 4; k++) sum += a[k][j] * b[i][k];
      dst[i][j] = sum;
    }
}

void transform_point(double out[3], Mat4 m, const double v[3]) { // Matrix::MultV, matrix.cc:209-216
  for (int c = 0; c < 3; c++) out[c] = m[0][c] * v[0] + m[1][c] * v[1] + m[2][c] * v[2] + m[3][c];
}

void rotation_from_quat(Mat4 m, const double q[4]) { // trackball.cc:272-291
  const double xx = q[0] * q[0], yy = q[1] * q[1], zz = q[2] * q[2];
  m[0][0] = 1.0 - 2.0 * (yy + zz);
  m[0][1] = 2.0 * (q[0] * q[1] - q[2] * q[3]);
  m[0][2] = 2.0 * (q[2] * q[0] + q[1] * q[3]);
  m[1][0] = 2.0 * (q[0] * q[1] + q[2] * q[3]);
  m[1][1] = 1.0 - 2.0 * (zz + xx);
  m[1][2] = 2.0 * (q[1] * q[2] - q[0] * q[3]);
  m[2][0] = 2.0 * (q[2] * q[0] - q[1] * q[3]);
  m[2][1] = 2.0 * (q[1] * q[2] + q[0] * q[3]);
  m[2][2] = 1.0 - 2.0 * (yy + xx);
  for (int k = 0; k < 3; k++) m[k][3] = m[3][k] = 0.0;
  m[3][3] = 1.0;
}

} // namespace

namespace mallie {

Camera::Camera(const double eye[3], const double lookat[3], const double up[3]) : fov_(45.0), height_(0), width_(0) {
  for (int k = 0; k < 3; k++) {
    eye_[k] = eye[k];
    up_[k] = up[k];
    lookat_[k] = lookat[k];
    origin_[k] = corner_[k] = du_[k] = dv_[k] = 0.0;
  }
}

void Camera::BuildCameraFrame(double origin[3], double corner[3], double u[3], double v[3], double fov,
                              const double quat[4], int width, int height) {
  width_ = width;
  height_ = height;
  Mat4 rot, view, m;
  rotation_from_quat(rot, quat);
  const double to_target[3] = {lookat_[0] - eye_[0], lookat_[1] - eye_[1], lookat_[2] - eye_[2]};
  const double dist = guarded_length(to_target);
  double pivot[3] = {0.0, 0.0, dist};
  invert(rot);
  const double zero[3] = {0.0, 0.0, 0.0}, y_up[3] = {0.0, 1.0, 0.0};
  look_at(view, pivot, zero, y_up);
  view[3][0] += eye_[0];
  view[3][1] += eye_[1];
  view[3][2] += (eye_[2] - dist);
  multiply(m, rot, view);
  double eye1[3], lookat1[3];
  transform_point(eye1, m, zero);
  pivot[2] = -pivot[2];
  transform_point(lookat1, m, pivot);
  const double *up1 = up_; // the caller's up vector is used untransformed (camera.cc:141-144)

  const double flen = (0.5f * (double)height / tanf(0.5f * (double)(fov * M_PI / 180.0f)));
  double look1[3] = {lookat1[0] - eye1[0], lookat1[1] - eye1[1], lookat1[2] - eye1[2]};
  cross(u, look1, up1);
  unitize<float>(u);
  cross(v, look1, u);
  unitize<float>(v);
  unitize<float>(look1);
  for (int k = 0; k < 3; k++) look1[k] = flen * look1[k] + eye1[k];
  for (int k = 0; k < 3; k++) corner[k] = look1[k] - 0.5f * (width * u[k] + height * v[k]);
  for (int k = 0; k < 3; k++) {
    origin[k] = eye1[k];
    origin_[k] = origin[k];
    corner_[k] = corner[k];
    du_[k] = u[k];
    dv_[k] = v[k];
  }
  fov_ = fov;
}

Ray Camera::GenerateRay(double u, double v) const {
  Ray ray;
  memset(&ray, 0, sizeof(ray));
  real3 d;
  d[0] = (corner_[0] + u * du_[0] + v * dv_[0]) - origin_[0];
  d[1] = (corner_[1] + u * du_[1] + v * dv_[1]) - origin_[1];
  d[2] = (corner_[2] + u * du_[2] + v * dv_[2]) - origin_[2];
  d.normalize();
  ray.org = real3(origin_[0], origin_[1], origin_[2]);
  ray.dir = d;
  return ray;
}

// camera.cc:242-257
Ray Camera::GenerateEnvRay(double u, double v) const {
  const double theta = M_PI * (v / height_);
  const double phi = 2.0 * M_PI * (u / width_);
  Ray ray;
  memset(&ray, 0, sizeof(ray));
  ray.org = real3(origin_[0], origin_[1], origin_[2]);
  ray.dir[0] = sin(theta) * cos(phi); // y up
  ray.dir[1] = cos(theta);
  ray.dir[2] = sin(theta) * sin(phi);
  return ray;
}

// camera.cc:259-329: top half of the frame = left eye, bottom half = right eye; the eyes sit on a circle of radius 0.5
// around origin_ and are toed in towards a focal distance of 4
Ray Camera::GenerateStereoEnvRay(double u, double v) const {
  const bool is_left_side = v < (height_ >> 1);
  const double focal_length = 4.0;
  const double r = 0.5;
  const double theta = M_PI * fmod(2.0 * v / height_, 1.0);
  const double phi = 2.0 * M_PI * (u / width_);
  real3 d0;
  d0[0] = sin(theta) * cos(phi);
  d0[1] = cos(theta);
  d0[2] = sin(theta) * sin(phi);
  real3 parallax;
  if (is_left_side) { // positive rotation
    parallax[0] = -d0[2];
    parallax[1] = 0.0;
    parallax[2] = d0[0];
  } else { // negative rotation
    parallax[0] = d0[2];
    parallax[1] = 0.0;
    parallax[2] = -d0[0];
  }
  parallax.normalize();
  parallax = parallax * r;
  Ray ray;
  memset(&ray, 0, sizeof(ray));
  ray.org[0] = origin_[0] + parallax[0];
  ray.org[1] = origin_[1] + parallax[1];
  ray.org[2] = origin_[2] + parallax[2];
  double psi = atan2(r, focal_length);
  if (is_left_side) psi = -psi;
  ray.dir[0] = d0[0] * cos(psi) - d0[2] * sin(psi);
  ray.dir[1] = d0[1];
  ray.dir[2] = d0[0] * sin(psi) + d0[2] * cos(psi);
  ray.dir.normalize();
  return ray;
}

// prim-plane.cc:8-44: the plane test in float, the record in double; faceID, u, v are left as they were
bool Plane::intersect(Intersection *info, const Ray &ray) {
  real3 n(m_a, m_b, m_c);
  real3 v = ray.dir;
  v.normalize();
  const float vn = (float)vdot(v, n);
  if (std::fabs(vn) > 1.1920928955078125e-07f * 1024.0f) { // std::numeric_limits<float>::epsilon() * 1024
    const float on_d = (float)(vdot(ray.org, n) + m_d);
    const float t = -on_d / vn;
    if ((t > 0) && (t < info->t)) {
      info->t = t;
      info->position = ray.org + (real)t * v;
      n.normalize();
      info->geometricNormal = n;
      info->normal = n;
      info->tangent[0] = 1.0;
      info->tangent[1] = 0.0;
      info->tangent[2] = 0.0;
      info->binormal[0] = 0.0;
      info->binormal[1] = 0.0;
      info->binormal[2] = -1.0;
      info->texcoord[0] = 0.0;
      info->texcoord[1] = 0.0;
      info->materialID = (unsigned int)(-1);
      return true;
    }
    return false;
  }
  return false;
}

} // namespace mallie

extern "C" int mgpu_camera_frame(const double eye[3], const double lookat[3], const double up[3], const double quat[4],
                                 double fov, int width, int height, double frame[12]) {
  if (!eye || !lookat || !up || !quat || !frame || width <= 0 || height <= 0) return MGPU_ERR_INVALID;
  mallie::Camera cam(eye, lookat, up);
  cam.BuildCameraFrame(frame + 0, frame + 3, frame + 6, frame + 9, fov, quat, width, height);
  return MGPU_OK;
}
