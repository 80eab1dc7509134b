// Host-side single-observation evaluator of the 9-parameter BAL camera behind the reference's
// BAConstraint surface (/root/reference/src/BAConstraint.cpp:10-86).  Block order and sizes are
// the reference's (point 3, rot 3, tr 3, intrinsics 6 -> 2 residuals, :16,:83); the algebra is
// the BAL model the north star names (rotate, translate, -P.xy/P.z, radial distortion, f), i.e.
// the reference's own commented BAL form (:89-128) with the sign and distortion applied.
// Residual sign: predicted - observed, the same as libmiba's kernels.
#include <iostream>

#include <ceres/rotation.h>
#include <Eigen/Dense>

#include "bundleadjust/BAConstraint.h"

BAConstraint::BAConstraint(const Eigen::Vector3f &observation, const float *estimatedPose) :
        observation{observation} {
    this->pose = Eigen::Map<const Eigen::Matrix4f>(estimatedPose);
}

template<typename T>
bool BAConstraint::operator()(const T *const point, const T *const rot, const T *const tr, const T *const intrinsics,
                              T *residuals) const {
    const T &fx = intrinsics[0];
    const T &fy = intrinsics[1];
    const T &ox = intrinsics[2];
    const T &oy = intrinsics[3];
    const T &k1 = intrinsics[4];
    const T &k2 = intrinsics[5];

    T p[3];
    ceres::AngleAxisRotatePoint(rot, point, p);
    p[0] += tr[0];
    p[1] += tr[1];
    p[2] += tr[2];

    // BAL cameras look down -z
    const T xp = -p[0] / p[2];
    const T yp = -p[1] / p[2];
    const T r2 = xp * xp + yp * yp;
    const T d = T(1.0) + r2 * (k1 + k2 * r2);

    residuals[0] = fx * d * xp + ox - T(double(observation(0)));
    residuals[1] = fy * d * yp + oy - T(double(observation(1)));
    return true;
}

template bool BAConstraint::operator()<double>(const double *, const double *, const double *, const double *, double *) const;
template bool BAConstraint::operator()<ceres::Jet<double, 15>>(const ceres::Jet<double, 15> *, const ceres::Jet<double, 15> *,
                                                                 const ceres::Jet<double, 15> *, const ceres::Jet<double, 15> *,
                                                                 ceres::Jet<double, 15> *) const;

ceres::CostFunction *BAConstraint::create(const Eigen::Vector3f &observation, const float *estimatedPose) {
    return new ceres::AutoDiffCostFunction<BAConstraint, 2, 3, 3, 3, 6>(
            new BAConstraint(observation, estimatedPose)
    );
}

bool BAConstraint::printOp(double *point, double *rot, double *tr, double *intrinsics, const float *) const {
    double res[2];
    (*this)(point, rot, tr, intrinsics, res);
    std::cout << "Proj: " << res[0] + observation(0) << " " << res[1] + observation(1) << "\n";
    std::cout << "Obs: " << observation(0) << " " << observation(1) << "\n";
    return true;
}
