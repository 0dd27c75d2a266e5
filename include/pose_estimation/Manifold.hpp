// pose_estimation/Manifold.hpp -- host value types with the MTK / ukfom NAMES the reference's public headers expose
// to callers:
//   MTK::SO3<double> (exp, log, boxplus, boxminus; src/pose_with_velocity/PoseUKF.cpp:135, BodyStateMeasurement.hpp:17),
//   MTK::vect<N>, ukfom::mtkwrap<M> (operator+ = copy [+], operator- = [-]; src/UnscentedKalmanFilter.hpp:23,
//   PoseWithVelocity.hpp:14-16), ukfom::ukf<WState>::cov / scalar_type / state (src/UnscentedKalmanFilter.hpp:24-25),
//   MTK::subblock / MTK::setDiagonal (src/pose_with_velocity/PoseUKF.cpp:104-107,184-185).
// These are CALLER CONVENIENCES on single values (a Rock component composing an initial state, converting a compass
// reading, comparing two states).  They are not the filter: ukfom::ukf here has no predict / update -- that arithmetic runs
// in the HIP kernels behind include/ukf_batch.h -- and nothing in this header is used by the engine or by oracle/.
// The maps follow MTK as recalled in SURVEY.md Appendix A.1 / A.2 (MTK is not vendored with the reference: unpinned; the
// tests check them against scipy's Rotation and closed forms, tests/test_abi_and_host.py):
//   exp(v, s)    = (cos(|v| s / 2), sinc(|v| s / 2) (s / 2) v), Taylor pairs below eps^(1/4)
//   log(q)       = 2 atan(|vec| / w) / |vec| * vec   (|vec| clamped from below at 1e-11; q and -q give the same vector)
//   q [+] v      = q * exp(v)   (right multiplication),   a [-] b = log(b^-1 * a)
//   vect: x [+] v = x + s v,  a [-] b = a - b;   compound manifolds: field by field in declaration order.
// With real MTK installed (<mtk/types/SOn.hpp> on the include path) this header defines nothing: the real types are used.
#ifndef POSE_ESTIMATION_MANIFOLD_HPP
#define POSE_ESTIMATION_MANIFOLD_HPP

#include <pose_estimation/Types.hpp>

#include <cmath>
#include <limits>

#if !(defined(__has_include) && __has_include(<mtk/types/SOn.hpp>))

namespace MTK {

// Euclidean vector as a manifold (mtk/types/vect.hpp): the matrix type itself plus DOF, boxplus, boxminus.
template <int N, typename Scalar = double> struct vect : pose_estimation::Matrix<Scalar, N, 1> {
    typedef pose_estimation::Matrix<Scalar, N, 1> base;
    enum { DOF = N };
    typedef Scalar scalar;
    vect() : base(base::Zero()) {}
    vect(const base& v) : base(v) {}
    template <typename Other> vect& operator=(const Other& v) { base::operator=(v); return *this; }
    void boxplus(const base& v, Scalar scale = Scalar(1)) { for (int k = 0; k < N; ++k) (*this)[k] += scale * v[k]; }
    void boxminus(base& res, const vect& other) const { for (int k = 0; k < N; ++k) res[k] = (*this)[k] - other[k]; }
};

// Unit quaternion as a 3-DOF manifold (mtk/types/SOn.hpp: derives from Eigen::Quaternion, coefficient order x y z w).
template <typename Scalar = double> struct SO3 : pose_estimation::Quaterniond {
    typedef pose_estimation::Quaterniond base;
    enum { DOF = 3 };
    typedef Scalar scalar;
    typedef vect<3, Scalar> vect_type;
    typedef pose_estimation::Matrix<Scalar, 3, 1> vec3;
    SO3() : base(base::Identity()) {}
    SO3(const base& q) : base(q) {}
    SO3(Scalar w, Scalar x, Scalar y, Scalar z) : base(w, x, y, z) {}
    SO3& operator=(const base& q) { base::operator=(q); return *this; }

    // (cos(sqrt(x2)), sin(sqrt(x2)) / sqrt(x2)): three Taylor pairs below eps^(1/4), the library functions above
    static void cos_sinc_sqrt(Scalar x2, Scalar& c, Scalar& s)
    {
        const Scalar bound = std::sqrt(std::sqrt(std::numeric_limits<Scalar>::epsilon()));
        if (x2 >= bound) {
            const Scalar x = std::sqrt(x2);
            c = std::cos(x);
            s = std::sin(x) / x;
            return;
        }
        Scalar cosi = Scalar(1), sinc = Scalar(1), term = Scalar(-0.5) * x2;
        cosi += term; term *= Scalar(1) / Scalar(3); sinc += term; term *= Scalar(-0.25) * x2;
        cosi += term; term *= Scalar(0.2);           sinc += term; term *= Scalar(-1) / Scalar(6) * x2;
        cosi += term; term *= Scalar(1) / Scalar(7); sinc += term;
        c = cosi;
        s = sinc;
    }
    /** rotation by the angle |vec| * scale about vec */
    static SO3 exp(const vec3& vec, Scalar scale = Scalar(1))
    {
        const Scalar half = Scalar(0.5) * scale;
        Scalar c, sc;
        cos_sinc_sqrt(half * half * vec.squaredNorm(), c, sc);
        const Scalar m = sc * half;
        return SO3(c, m * vec[0], m * vec[1], m * vec[2]);
    }
    /** rotation vector of orient (angle in (-pi, pi): orient and -orient give the same vector) */
    static vec3 log(const SO3& orient)
    {
        const Scalar vx = orient.x(), vy = orient.y(), vz = orient.z();
        Scalar nv = std::sqrt(vx * vx + vy * vy + vz * vz);
        const Scalar tol = (sizeof(Scalar) == 8) ? Scalar(1e-11) : Scalar(1e-5);
        if (nv < tol) nv = tol;
        const Scalar f = Scalar(2) / nv * std::atan(nv / orient.w());
        return vec3(f * vx, f * vy, f * vz);
    }
    /** MTK's spelling: the result is the first argument */
    static void log(vec3& res, const SO3& orient) { res = log(orient); }
    void boxplus(const vec3& vec, Scalar scale = Scalar(1)) { *this = SO3(static_cast<const base&>(*this) * exp(vec, scale)); }
    void boxminus(vec3& res, const SO3& other) const { res = log(SO3(other.conjugate() * static_cast<const base&>(*this))); }
};

}  // namespace MTK

namespace ukfom {

// ukfom/mtkwrap.hpp: the manifold plus value-returning operators.  x + delta = copy of x moved by delta, a - b = the tangent
// vector from b to a.  (Free operators here, members in ukfom: `a + d` and `a - b` read the same.)
template <typename M> struct mtkwrap : M {
    typedef mtkwrap<M> self;
    typedef M MTK_type;
    enum { DOF = M::DOF };
    typedef typename M::scalar scalar;
    typedef pose_estimation::Matrix<scalar, int(M::DOF), 1> vectorized_type;
    mtkwrap() : M() {}
    mtkwrap(const M& m) : M(m) {}
    template <typename A> explicit mtkwrap(const A& a) : M(a) {}
    template <typename A> self& operator=(const A& a) { M::operator=(a); return *this; }
};
template <typename M> mtkwrap<M> operator+(const mtkwrap<M>& x, const typename mtkwrap<M>::vectorized_type& delta)
{
    mtkwrap<M> r(x);
    r.boxplus(delta);
    return r;
}
template <typename M> typename mtkwrap<M>::vectorized_type operator-(const mtkwrap<M>& a, const mtkwrap<M>& b)
{
    typename mtkwrap<M>::vectorized_type r;
    a.boxminus(r, b);
    return r;
}

// ukfom/ukf.hpp: the TYPE names only.  predict() / update() / mu() / sigma() are the engine's (include/ukf_batch.h), reached
// through pose_estimation::UnscentedKalmanFilter; there is deliberately no CPU arithmetic to call here.
template <typename State> struct ukf {
    typedef State state;
    typedef typename State::scalar scalar_type;
    enum { n = State::DOF };
    typedef pose_estimation::Matrix<scalar_type, int(State::DOF), int(State::DOF)> cov;
    typedef pose_estimation::Matrix<scalar_type, int(State::DOF), 1> state_vector;
private:
    ukf();
};

// ukfom/util.hpp
template <typename T> inline const T& id(const T& x) { return x; }
template <typename scalar> inline bool accept_any_mahalanobis_distance(const scalar&) { return true; }

}  // namespace ukfom

namespace MTK {

// mtk/startIdx.hpp: the covariance block of one field of a compound manifold, named by its member pointer.  The compound
// manifolds of this package (PoseWithVelocity, OrientationState) tell the tangent offset of a field through
// Manifold::tangentIndex(member pointer) / the field's own DOF.
template <typename Scalar, int D, typename Field, typename Manifold>
pose_estimation::FixedBlock<pose_estimation::Matrix<Scalar, D, D>, Scalar, int(Field::DOF), int(Field::DOF)>
subblock(pose_estimation::Matrix<Scalar, D, D>& cov, Field Manifold::*field)
{
    const int i = Manifold::tangentIndex(field);
    return pose_estimation::FixedBlock<pose_estimation::Matrix<Scalar, D, D>, Scalar, int(Field::DOF), int(Field::DOF)>(cov, i, i);
}
template <typename Scalar, int D, typename Field, typename Manifold>
pose_estimation::FixedBlock<const pose_estimation::Matrix<Scalar, D, D>, Scalar, int(Field::DOF), int(Field::DOF)>
subblock(const pose_estimation::Matrix<Scalar, D, D>& cov, Field Manifold::*field)
{
    const int i = Manifold::tangentIndex(field);
    return pose_estimation::FixedBlock<const pose_estimation::Matrix<Scalar, D, D>, Scalar, int(Field::DOF), int(Field::DOF)>(cov, i, i);
}
/** the field's diagonal block becomes val * identity */
template <typename Scalar, int D, typename Field, typename Manifold>
void setDiagonal(pose_estimation::Matrix<Scalar, D, D>& cov, Field Manifold::*field, Scalar val)
{
    const int i = Manifold::tangentIndex(field);
    for (int r = 0; r < int(Field::DOF); ++r)
        for (int c = 0; c < int(Field::DOF); ++c) cov(i + r, i + c) = (r == c) ? val : Scalar(0);
}

}  // namespace MTK

#endif  // no real MTK
#endif
