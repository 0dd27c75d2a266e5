// pose_estimation/Measurement.hpp -- same macro and member names as the reference's
// src/Measurement.hpp:6-16 ({Mu mu = 0; Cov cov = Identity}), on the dependency-free matrix type.
#ifndef _POSE_ESTIMATION_MEASUREMENT_HPP
#define _POSE_ESTIMATION_MEASUREMENT_HPP

#include <pose_estimation/Types.hpp>

#define MEASUREMENT(NAME, DIM) \
struct NAME \
{ \
typedef pose_estimation::Matrix<double, DIM, 1> Mu; \
typedef pose_estimation::Matrix<double, DIM, DIM> Cov; \
 \
NAME() : mu(Mu::Zero()), cov(Cov::Identity()) {} \
 \
Mu mu; \
Cov cov; \
};

#endif
