// pose_estimation/Types.hpp -- dependency-free fixed-size matrix / quaternion types standing in for
// Eigen and MTK at the HOST boundary of the MI355X engine (Eigen, boost and MTK are absent from this
// image).  They carry values only; all filter arithmetic happens in the HIP kernels behind
// include/ukf_batch.h.  Storage is row-major; symmetric matrices make that indistinguishable from
// Eigen's column-major default at this boundary.
#ifndef POSE_ESTIMATION_TYPES_HPP
#define POSE_ESTIMATION_TYPES_HPP

#include <cmath>
#include <cstddef>

namespace pose_estimation {

template <typename Scalar, int R, int C> struct Matrix {
    enum { Rows = R, Cols = C };
    Scalar d[R * C];
    Matrix() { for (int i = 0; i < R * C; ++i) d[i] = Scalar(0); }
    static Matrix Zero() { return Matrix(); }
    static Matrix Ones() { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = Scalar(1); return m; }
    static Matrix Identity() { Matrix m; for (int i = 0; i < (R < C ? R : C); ++i) m.d[i * C + i] = Scalar(1); return m; }
    Scalar& operator()(int r, int c) { return d[r * C + c]; }
    const Scalar& operator()(int r, int c) const { return d[r * C + c]; }
    Scalar& operator()(int i) { return d[i]; }
    const Scalar& operator()(int i) const { return d[i]; }
    Scalar& operator[](int i) { return d[i]; }
    const Scalar& operator[](int i) const { return d[i]; }
    Scalar& x() { return d[0]; } Scalar& y() { return d[1]; } Scalar& z() { return d[2]; }
    const Scalar& x() const { return d[0]; } const Scalar& y() const { return d[1]; } const Scalar& z() const { return d[2]; }
    const Scalar* data() const { return d; }
    Scalar* data() { return d; }
    bool allFinite() const { for (int i = 0; i < R * C; ++i) if (!std::isfinite(d[i])) return false; return true; }
    Matrix operator*(Scalar s) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] * s; return m; }
    Matrix operator-(const Matrix& o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] - o.d[i]; return m; }
    Matrix operator+(const Matrix& o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] + o.d[i]; return m; }
    // block(r0, c0) of size BRxBC, copy in / out
    template <int BR, int BC> Matrix<Scalar, BR, BC> block(int r0, int c0) const {
        Matrix<Scalar, BR, BC> b;
        for (int r = 0; r < BR; ++r) for (int c = 0; c < BC; ++c) b(r, c) = (*this)(r0 + r, c0 + c);
        return b;
    }
    template <int BR, int BC> void setBlock(int r0, int c0, const Matrix<Scalar, BR, BC>& b) {
        for (int r = 0; r < BR; ++r) for (int c = 0; c < BC; ++c) (*this)(r0 + r, c0 + c) = b(r, c);
    }
};
template <typename Scalar, int R, int C> Matrix<Scalar, R, C> operator*(Scalar s, const Matrix<Scalar, R, C>& m) { return m * s; }

typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 3, 3> Matrix3d;

// Unit quaternion in Eigen coefficient order (x, y, z, w); MTK::SO3<double> derives from it.
struct Quaterniond {
    double c[4];
    Quaterniond() { c[0] = c[1] = c[2] = 0.0; c[3] = 1.0; }
    Quaterniond(double w, double x, double y, double z) { c[0] = x; c[1] = y; c[2] = z; c[3] = w; }
    static Quaterniond Identity() { return Quaterniond(); }
    double& x() { return c[0]; } double& y() { return c[1]; } double& z() { return c[2]; } double& w() { return c[3]; }
    const double& x() const { return c[0]; } const double& y() const { return c[1]; }
    const double& z() const { return c[2]; } const double& w() const { return c[3]; }
    const double* coeffs() const { return c; }
    double* coeffs() { return c; }
    // rotation of a vector (Eigen _transformVector); a host convenience, not used by the filter path
    Vector3d operator*(const Vector3d& v) const {
        double ux = c[1] * v[2] - c[2] * v[1], uy = c[2] * v[0] - c[0] * v[2], uz = c[0] * v[1] - c[1] * v[0];
        ux += ux; uy += uy; uz += uz;
        Vector3d r;
        r[0] = v[0] + c[3] * ux + (c[1] * uz - c[2] * uy);
        r[1] = v[1] + c[3] * uy + (c[2] * ux - c[0] * uz);
        r[2] = v[2] + c[3] * uz + (c[0] * uy - c[1] * ux);
        return r;
    }
};

}  // namespace pose_estimation

#endif
