// pose_estimation/Types.hpp -- the value types of the HOST boundary of the MI355X engine.
//
// The reference's public headers and its callers are written against Eigen and MTK
// (`Eigen::Matrix<double, DIM, 1>` in src/Measurement.hpp:9-10 and src/UnscentedKalmanFilter.hpp:142-147,
// `filter_state_cov.block(0, 0, 3, 3) = ...; filter_state_cov.setZero();` in
// src/pose_with_velocity/BodyStateMeasurement.hpp:21-25,35-38, `MTK::SO3<double>(q)` in :17).  Where Eigen is
// installed these types CAN BE Eigen's (opt in: -DPOSE_ESTIMATION_USE_EIGEN, see below); by default, and in an image
// without Eigen (this one), the dependency-free stand-ins below offer the members that code uses, with Eigen's names
// and semantics, so the same caller text compiles either way.  They carry values only; all filter arithmetic happens in the HIP kernels
// behind include/ukf_batch.h.  The stand-in stores row-major, Eigen column-major: matrices therefore cross the C-ABI
// (row-major, lower triangle read -- as Eigen's LLT reads the reference's sigma_) through the explicit (row, column)
// loops of to_row_major / from_row_major below, never through data().
//
// THE REAL-EIGEN BRANCH HAS NEVER BEEN COMPILED in the build image (Eigen is absent there): it is untested, and therefore
// not the default -- the stand-ins are what every test of this repository has run.  -DPOSE_ESTIMATION_USE_EIGEN selects it
// where Eigen is installed.  Two differences to expect: a default-constructed Eigen matrix is
// uninitialised where the stand-in zero-fills, and Eigen's expression templates may need .eval() where the stand-in
// returns values.  (-DPOSE_ESTIMATION_NO_EIGEN, the switch of earlier rounds, still forces the stand-ins.)
#ifndef POSE_ESTIMATION_TYPES_HPP
#define POSE_ESTIMATION_TYPES_HPP

#include <cmath>
#include <cstddef>

#if defined(__has_include)
#if defined(POSE_ESTIMATION_USE_EIGEN) && __has_include(<Eigen/Core>) && __has_include(<Eigen/Geometry>) && !defined(POSE_ESTIMATION_NO_EIGEN)
#define POSE_ESTIMATION_HAS_EIGEN 1
#endif
#endif

#ifdef POSE_ESTIMATION_HAS_EIGEN
// ------------------------------------------------------------------------------------------------ real Eigen
#include <Eigen/Core>
#include <Eigen/Geometry>

namespace pose_estimation {
template <typename Scalar, int R, int C> using Matrix = Eigen::Matrix<Scalar, R, C>;
typedef Eigen::Vector3d Vector3d;
typedef Eigen::Matrix3d Matrix3d;
typedef Eigen::Quaterniond Quaterniond;
template <typename M, typename Scalar, int BR, int BC> using FixedBlock = Eigen::Block<M, BR, BC>;   // (xpr, startRow, startCol)
}  // namespace pose_estimation

#else
// ------------------------------------------------------------------------------------------------ stand-ins
namespace pose_estimation {

template <typename Scalar, int R, int C> struct Matrix;

// Assignable view of a rectangular part of a matrix: what Eigen's block(r, c, h, w) returns.
template <typename M, typename Scalar> class BlockView {
public:
    BlockView(M& m, int r0, int c0, int h, int w) : m_(m), r0_(r0), c0_(c0), h_(h), w_(w) {}
    int rows() const { return h_; }
    int cols() const { return w_; }
    Scalar& operator()(int r, int c) { return m_(r0_ + r, c0_ + c); }
    const Scalar& operator()(int r, int c) const { return m_(r0_ + r, c0_ + c); }
    template <int BR, int BC> BlockView& operator=(const Matrix<Scalar, BR, BC>& b) {
        for (int r = 0; r < h_; ++r) for (int c = 0; c < w_; ++c) m_(r0_ + r, c0_ + c) = b(r, c);
        return *this;
    }
    template <typename M2> BlockView& operator=(const BlockView<M2, Scalar>& b) {
        for (int r = 0; r < h_; ++r) for (int c = 0; c < w_; ++c) m_(r0_ + r, c0_ + c) = b(r, c);
        return *this;
    }
    BlockView& operator=(const BlockView& b) { return this->template operator=<M>(b); }
    template <int BR, int BC> BlockView& operator+=(const Matrix<Scalar, BR, BC>& b) {
        for (int r = 0; r < h_; ++r) for (int c = 0; c < w_; ++c) m_(r0_ + r, c0_ + c) += b(r, c);
        return *this;
    }
    BlockView& setZero() { for (int r = 0; r < h_; ++r) for (int c = 0; c < w_; ++c) m_(r0_ + r, c0_ + c) = Scalar(0); return *this; }
    BlockView& setIdentity() { for (int r = 0; r < h_; ++r) for (int c = 0; c < w_; ++c) m_(r0_ + r, c0_ + c) = Scalar(r == c); return *this; }
    BlockView& operator*=(Scalar s) { for (int r = 0; r < h_; ++r) for (int c = 0; c < w_; ++c) m_(r0_ + r, c0_ + c) *= s; return *this; }
    bool allFinite() const { for (int r = 0; r < h_; ++r) for (int c = 0; c < w_; ++c) if (!std::isfinite((*this)(r, c))) return false; return true; }

private:
    M& m_;
    int r0_, c0_, h_, w_;
};

// The same with the size known at compile time: what Eigen's block<BR, BC>(r, c) returns (and MTK::subblock).  It takes part in
// products like a BR x BC matrix.
template <typename M, typename Scalar, int BR, int BC> class FixedBlock : public BlockView<M, Scalar> {
public:
    FixedBlock(M& m, int r0, int c0) : BlockView<M, Scalar>(m, r0, c0, BR, BC) {}
    using BlockView<M, Scalar>::operator=;
    FixedBlock& operator=(const FixedBlock& b) { BlockView<M, Scalar>::operator=(b); return *this; }
    Matrix<Scalar, BR, BC> eval() const { return Matrix<Scalar, BR, BC>(*this); }
    template <int K> Matrix<Scalar, BR, K> operator*(const Matrix<Scalar, BC, K>& o) const { return eval() * o; }
    Matrix<Scalar, BC, BR> transpose() const { return eval().transpose(); }
};

template <typename Scalar, int R, int C> struct Matrix {
    enum { Rows = R, Cols = C, RowsAtCompileTime = R, ColsAtCompileTime = C, SizeAtCompileTime = R * C };
    typedef Scalar value_type;
    Scalar d[R * C];
    Matrix() { for (int i = 0; i < R * C; ++i) d[i] = Scalar(0); }
    // fixed-size vector constructors, as Eigen's
    Matrix(Scalar a, Scalar b) { static_assert(R * C == 2, "2-vector"); d[0] = a; d[1] = b; }
    Matrix(Scalar a, Scalar b, Scalar c) { static_assert(R * C == 3, "3-vector"); d[0] = a; d[1] = b; d[2] = c; }
    // a block of another matrix converts to a matrix of the same size (Eigen evaluates the expression)
    template <typename M2> Matrix(const BlockView<M2, Scalar>& b) { for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) (*this)(r, c) = b(r, c); }
    template <typename M2> Matrix& operator=(const BlockView<M2, Scalar>& b) {
        for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) (*this)(r, c) = b(r, c);
        return *this;
    }
    static Matrix Zero() { return Matrix(); }
    static Matrix Ones() { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = Scalar(1); return m; }
    static Matrix Constant(Scalar v) { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = v; return m; }
    static Matrix Identity() { Matrix m; for (int i = 0; i < (R < C ? R : C); ++i) m.d[i * C + i] = Scalar(1); return m; }
    Matrix& setZero() { for (int i = 0; i < R * C; ++i) d[i] = Scalar(0); return *this; }
    Matrix& setOnes() { for (int i = 0; i < R * C; ++i) d[i] = Scalar(1); return *this; }
    Matrix& setConstant(Scalar v) { for (int i = 0; i < R * C; ++i) d[i] = v; return *this; }
    Matrix& setIdentity() { setZero(); for (int i = 0; i < (R < C ? R : C); ++i) d[i * C + i] = Scalar(1); return *this; }
    int rows() const { return R; }
    int cols() const { return C; }
    int size() const { return R * C; }
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
    bool hasNaN() const { for (int i = 0; i < R * C; ++i) if (d[i] != d[i]) return true; return false; }
    Scalar squaredNorm() const { Scalar s = 0; for (int i = 0; i < R * C; ++i) s += d[i] * d[i]; return s; }
    Scalar norm() const { return std::sqrt(squaredNorm()); }
    Scalar sum() const { Scalar s = 0; for (int i = 0; i < R * C; ++i) s += d[i]; return s; }
    Scalar trace() const { Scalar s = 0; for (int i = 0; i < (R < C ? R : C); ++i) s += d[i * C + i]; return s; }
    Scalar dot(const Matrix& o) const { Scalar s = 0; for (int i = 0; i < R * C; ++i) s += d[i] * o.d[i]; return s; }
    Matrix cross(const Matrix& o) const {
        static_assert(R * C == 3, "cross product of 3-vectors");
        return Matrix(d[1] * o.d[2] - d[2] * o.d[1], d[2] * o.d[0] - d[0] * o.d[2], d[0] * o.d[1] - d[1] * o.d[0]);
    }
    Matrix<Scalar, C, R> transpose() const { Matrix<Scalar, C, R> t; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) t(c, r) = (*this)(r, c); return t; }
    Matrix operator-() const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = -d[i]; return m; }
    Matrix operator*(Scalar s) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] * s; return m; }
    Matrix operator/(Scalar s) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] / s; return m; }
    Matrix operator-(const Matrix& o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] - o.d[i]; return m; }
    Matrix operator+(const Matrix& o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] + o.d[i]; return m; }
    Matrix& operator+=(const Matrix& o) { for (int i = 0; i < R * C; ++i) d[i] += o.d[i]; return *this; }
    Matrix& operator-=(const Matrix& o) { for (int i = 0; i < R * C; ++i) d[i] -= o.d[i]; return *this; }
    Matrix& operator*=(Scalar s) { for (int i = 0; i < R * C; ++i) d[i] *= s; return *this; }
    bool operator==(const Matrix& o) const { for (int i = 0; i < R * C; ++i) if (!(d[i] == o.d[i])) return false; return true; }
    bool operator!=(const Matrix& o) const { return !(*this == o); }
    template <int K> Matrix<Scalar, R, K> operator*(const Matrix<Scalar, C, K>& o) const {
        Matrix<Scalar, R, K> m;
        for (int r = 0; r < R; ++r) for (int k = 0; k < K; ++k) { Scalar s = 0; for (int c = 0; c < C; ++c) s += (*this)(r, c) * o(c, k); m(r, k) = s; }
        return m;
    }
    // Eigen's two block forms: run-time size (a view) and compile-time size (also a view)
    BlockView<Matrix, Scalar> block(int r0, int c0, int h, int w) { return BlockView<Matrix, Scalar>(*this, r0, c0, h, w); }
    BlockView<const Matrix, Scalar> block(int r0, int c0, int h, int w) const { return BlockView<const Matrix, Scalar>(*this, r0, c0, h, w); }
    template <int BR, int BC> FixedBlock<Matrix, Scalar, BR, BC> block(int r0, int c0) { return FixedBlock<Matrix, Scalar, BR, BC>(*this, r0, c0); }
    template <int BR, int BC> FixedBlock<const Matrix, Scalar, BR, BC> block(int r0, int c0) const { return FixedBlock<const Matrix, Scalar, BR, BC>(*this, r0, c0); }
    template <typename M2, int K> Matrix<Scalar, R, K> operator*(const FixedBlock<M2, Scalar, C, K>& b) const { return *this * b.eval(); }
    BlockView<Matrix, Scalar> topLeftCorner(int h, int w) { return block(0, 0, h, w); }
    BlockView<Matrix, Scalar> row(int r) { return block(r, 0, 1, C); }
    BlockView<Matrix, Scalar> col(int c) { return block(0, c, R, 1); }
    template <int N> BlockView<Matrix, Scalar> head() { return C == 1 ? block(0, 0, N, 1) : block(0, 0, 1, N); }
    template <int N> BlockView<Matrix, Scalar> segment(int i) { return C == 1 ? block(i, 0, N, 1) : block(0, i, 1, N); }
};
template <typename Scalar, int R, int C> Matrix<Scalar, R, C> operator*(Scalar s, const Matrix<Scalar, R, C>& m) { return m * s; }

typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 3, 3> Matrix3d;

// Unit quaternion in Eigen coefficient order (x, y, z, w); MTK::SO3<double> derives from Eigen::Quaterniond.
struct Quaterniond {
    double c[4];
    Quaterniond() { c[0] = c[1] = c[2] = 0.0; c[3] = 1.0; }
    Quaterniond(double w, double x, double y, double z) { c[0] = x; c[1] = y; c[2] = z; c[3] = w; }
    static Quaterniond Identity() { return Quaterniond(); }
    Quaterniond& setIdentity() { c[0] = c[1] = c[2] = 0.0; c[3] = 1.0; return *this; }
    double& x() { return c[0]; } double& y() { return c[1]; } double& z() { return c[2]; } double& w() { return c[3]; }
    const double& x() const { return c[0]; } const double& y() const { return c[1]; }
    const double& z() const { return c[2]; } const double& w() const { return c[3]; }
    const double* coeffs() const { return c; }
    double* coeffs() { return c; }
    Vector3d vec() const { return Vector3d(c[0], c[1], c[2]); }
    double squaredNorm() const { return c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]; }
    double norm() const { return std::sqrt(squaredNorm()); }
    void normalize() { const double n = norm(); for (int k = 0; k < 4; ++k) c[k] /= n; }
    Quaterniond normalized() const { Quaterniond q(*this); q.normalize(); return q; }
    Quaterniond conjugate() const { return Quaterniond(c[3], -c[0], -c[1], -c[2]); }
    Quaterniond inverse() const { const double n2 = squaredNorm(); return Quaterniond(c[3] / n2, -c[0] / n2, -c[1] / n2, -c[2] / n2); }
    Quaterniond operator*(const Quaterniond& b) const {   // Hamilton product, Eigen's operand order
        return Quaterniond(c[3] * b.c[3] - c[0] * b.c[0] - c[1] * b.c[1] - c[2] * b.c[2],
                           c[3] * b.c[0] + c[0] * b.c[3] + c[1] * b.c[2] - c[2] * b.c[1],
                           c[3] * b.c[1] + c[1] * b.c[3] + c[2] * b.c[0] - c[0] * b.c[2],
                           c[3] * b.c[2] + c[2] * b.c[3] + c[0] * b.c[1] - c[1] * b.c[0]);
    }
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
    Matrix3d toRotationMatrix() const {   // Eigen toRotationMatrix
        const double tx = 2 * c[0], ty = 2 * c[1], tz = 2 * c[2];
        const double twx = tx * c[3], twy = ty * c[3], twz = tz * c[3];
        const double txx = tx * c[0], txy = ty * c[0], txz = tz * c[0], tyy = ty * c[1], tyz = tz * c[1], tzz = tz * c[2];
        Matrix3d m;
        m(0, 0) = 1 - (tyy + tzz); m(0, 1) = txy - twz;       m(0, 2) = txz + twy;
        m(1, 0) = txy + twz;       m(1, 1) = 1 - (txx + tzz); m(1, 2) = tyz - twx;
        m(2, 0) = txz - twy;       m(2, 1) = tyz + twx;       m(2, 2) = 1 - (txx + tyy);
        return m;
    }
    Matrix3d matrix() const { return toRotationMatrix(); }
};

}  // namespace pose_estimation

// the reference's headers spell the types with their Eigen names
namespace Eigen {
template <typename Scalar, int R, int C> using Matrix = pose_estimation::Matrix<Scalar, R, C>;
typedef pose_estimation::Vector3d Vector3d;
typedef pose_estimation::Matrix3d Matrix3d;
typedef pose_estimation::Quaterniond Quaterniond;
}  // namespace Eigen
#endif  // POSE_ESTIMATION_HAS_EIGEN

namespace pose_estimation {
// matrix <-> the row-major arrays of include/ukf_batch.h, entry by entry: correct for either storage order of Matrix and
// for a matrix that is not exactly symmetric (the engine reads the LOWER triangle, m(r, c) with c <= r)
template <typename Scalar, int R, int C> inline void to_row_major(const Matrix<Scalar, R, C>& m, double* out)
{
    for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) out[r * C + c] = m(r, c);
}
template <typename Scalar, int R, int C> inline void from_row_major(const double* in, Matrix<Scalar, R, C>& m)
{
    for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) m(r, c) = in[r * C + c];
}
}  // namespace pose_estimation

// the MTK / ukfom value types the reference's public headers name (MTK::SO3, MTK::vect, ukfom::mtkwrap, ukfom::ukf<>::cov)
#include <pose_estimation/Manifold.hpp>

#endif
