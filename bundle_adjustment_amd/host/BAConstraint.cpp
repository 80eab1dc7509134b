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

namespace { bool g_reference_model = false; }
void BAConstraint::useReferenceModel(bool on) { g_reference_model = on; }
bool BAConstraint::referenceModel() { return g_reference_model; }

template<typename T>
bool BAConstraint::operator()(const T *const point, const T *const rot, const T *const tr, const T *const intrinsics,
                              T *residuals) const {
    if (g_reference_model) {
        // the reference's functor, expression by expression (src/BAConstraint.cpp:34-77); the inverse of the 4x4 extrinsic
        // [R tr; 0 1] is [R^-1, -R^-1 tr; 0 1], so cam_point = R^-1 (point - tr) with R^-1 by cofactors
        T R[9];
        ceres::AngleAxisToRotationMatrix<T>(rot, R);   // column-major, as ceres
        const T a00 = R[0], a01 = R[3], a02 = R[6], a10 = R[1], a11 = R[4], a12 = R[7], a20 = R[2], a21 = R[5], a22 = R[8];
        const T c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
        const T id = T(1.0) / (a00 * c00 + a01 * c01 + a02 * c02);
        const T y0 = point[0] - tr[0], y1 = point[1] - tr[1], y2 = point[2] - tr[2];
        const T P0 = (c00 * y0 + (a02 * a21 - a01 * a22) * y1 + (a01 * a12 - a02 * a11) * y2) * id;
        const T P1 = (c01 * y0 + (a00 * a22 - a02 * a20) * y1 + (a02 * a10 - a00 * a12) * y2) * id;
        const T P2 = (c02 * y0 + (a01 * a20 - a00 * a21) * y1 + (a00 * a11 - a01 * a10) * y2) * id;
        const T u = P0 / P2, v = P1 / P2;              // no sign flip (:67-68); the radial factor is computed and dropped (:69-70)
        residuals[0] = T(double(observation(0))) - (intrinsics[0] * u + intrinsics[2]);
        residuals[1] = T(double(observation(1))) - (intrinsics[1] * v + intrinsics[3]);
        return true;
    }
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
