// Reprojection constraint, same surface as /root/reference/include/bundleadjust/BAConstraint.h:17-32.
// The functor evaluates the 9-parameter BAL camera (BASELINE.json north_star; the reference's own
// BAL form at src/BAConstraint.cpp:89-128) on the host for ONE observation; the solve path never
// allocates one of these per observation, it runs the same model in HIP (libmiba K1).
#pragma once

#include <ceres/cost_function.h>
#include <ceres/ceres.h>
#include <opencv2/opencv.hpp>
#include "bundleadjust/Dataloader.h"

#include "BalDataloader.h"

class BAConstraint {

public:
    BAConstraint(const Eigen::Vector3f& observation, const float * estimatedPose);

    // blocks: point[3], rot[3] (angle-axis), tr[3], intrinsics[6] = fx,fy,ox,oy,k1,k2; residuals[2]
    template <typename T>
    bool operator()(const T* const point, const T* const rot, const T* const tr, const T* const intrinsics, T* residuals) const;

    bool printOp(double * point, double * rot, double * tr, double * intrinsics, const float * estimatedPose) const;

    static ceres::CostFunction* create(const Eigen::Vector3f& observation, const float * estimatedPose);

protected:
    Eigen::Matrix4f pose;
    const Eigen::Vector3f observation;
};
