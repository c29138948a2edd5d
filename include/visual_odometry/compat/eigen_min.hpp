// Minimal Eigen::Matrix3d / Eigen::Vector3d stand-ins (column-major storage,
// operator()(r,c), products, transpose) for builds without Eigen.
#pragma once
#if __has_include(<Eigen/Core>) && !defined(VO_FORCE_COMPAT_TYPES)
#include <Eigen/Core>
#include <Eigen/Geometry>
#else
#include <cmath>
#define VO_COMPAT_EIGEN_TYPES 1
namespace Eigen {

struct Vector3d {
    double v[3]{0, 0, 0};
    Vector3d() = default;
    Vector3d(double x, double y, double z) : v{x, y, z} {}
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    [[nodiscard]] double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
    void normalize() { double n = norm(); if (n > 0) { v[0] /= n; v[1] /= n; v[2] /= n; } }
    Vector3d operator+(Vector3d const& o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
    Vector3d operator-(Vector3d const& o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
};

struct Matrix3d {
    double m[9]{0, 0, 0, 0, 0, 0, 0, 0, 0};   // column-major
    double& operator()(int r, int c) { return m[c * 3 + r]; }
    double operator()(int r, int c) const { return m[c * 3 + r]; }
    static Matrix3d Identity() { Matrix3d I; I(0, 0) = I(1, 1) = I(2, 2) = 1; return I; }
    Matrix3d operator*(Matrix3d const& o) const {
        Matrix3d r;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r(i, j) = (*this)(i, 0) * o(0, j) + (*this)(i, 1) * o(1, j) + (*this)(i, 2) * o(2, j);
        return r;
    }
    Vector3d operator*(Vector3d const& x) const {
        return {(*this)(0, 0) * x(0) + (*this)(0, 1) * x(1) + (*this)(0, 2) * x(2),
                (*this)(1, 0) * x(0) + (*this)(1, 1) * x(1) + (*this)(1, 2) * x(2),
                (*this)(2, 0) * x(0) + (*this)(2, 1) * x(1) + (*this)(2, 2) * x(2)};
    }
    [[nodiscard]] Matrix3d transpose() const {
        Matrix3d r;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r(i, j) = (*this)(j, i);
        return r;
    }
    [[nodiscard]] double determinant() const {
        auto const& a = *this;
        return a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) - a(0, 1) * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) +
               a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0));
    }
};

}  // namespace Eigen
#endif
