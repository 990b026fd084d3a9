// Minimal stand-ins for the Eigen / Sophus / OpenCV types that appear in the signatures of the
// reference's map classes (KeyFrame.h:370 Sophus::SE3f mTcw, MapPoint.h:179 Eigen::Vector3f,
// cv::KeyPoint).  Only what the Optimizer adapter touches; used for the compile check and the
// adapter test on machines without those libraries.  The real headers replace this directory
// when the adapter is built inside MoV-SLAM.
#pragma once
#include <cmath>

namespace Eigen {
struct Vector3f {
    float v[3];
    Vector3f() : v{0, 0, 0} {}
    Vector3f(float x, float y, float z) : v{x, y, z} {}
    float operator()(int i) const { return v[i]; }
    float &operator()(int i) { return v[i]; }
    float x() const { return v[0]; } float y() const { return v[1]; } float z() const { return v[2]; }
    // the float arithmetic MapPoint::UpdateNormalAndDepth uses (MapPoint.cc:380-431)
    Vector3f operator-(const Vector3f &o) const { return Vector3f(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vector3f operator+(const Vector3f &o) const { return Vector3f(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vector3f operator/(float s) const { return Vector3f(v[0] / s, v[1] / s, v[2] / s); }
    float norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
    void setZero() { v[0] = v[1] = v[2] = 0.f; }
};
struct Quaternionf {
    float qx, qy, qz, qw;
    Quaternionf() : qx(0), qy(0), qz(0), qw(1) {}
    Quaternionf(float w, float x, float y, float z) : qx(x), qy(y), qz(z), qw(w) {}      // Eigen order: w first
    float x() const { return qx; } float y() const { return qy; } float z() const { return qz; } float w() const { return qw; }
};
struct Matrix3d { double m[9]; };
}  // namespace Eigen

namespace Sophus {
// SE3f(q, t) normalises the quaternion like Sophus does
struct SE3f {
    Eigen::Quaternionf q; Eigen::Vector3f t;
    SE3f() {}
    SE3f(const Eigen::Quaternionf &q_, const Eigen::Vector3f &t_) : q(q_), t(t_) {
        const float n = std::sqrt(q.qx * q.qx + q.qy * q.qy + q.qz * q.qz + q.qw * q.qw);
        q.qx /= n; q.qy /= n; q.qz /= n; q.qw /= n;
    }
    const Eigen::Quaternionf &unit_quaternion() const { return q; }
    const Eigen::Vector3f &translation() const { return t; }
};
template <class T> using SE3 = SE3f;
}  // namespace Sophus

namespace cv {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; int octave = 0; };
}  // namespace cv
