// One reprojection constraint on the host.
//
// Public surface of the reference's class (/root/reference/include/bundleadjust/BAConstraint.h:17-32): constructor,
// templated functor, printOp, create.  What is behind it differs: the functor evaluates the 9-parameter BAL camera
// (BASELINE.json north_star; the reference keeps that form commented at src/BAConstraint.cpp:89-128), and the solve
// path never builds one of these per observation — libmiba runs the same model in HIP.  The class stays for
// callers that want the residual or its automatic derivative of a single observation (tests do).
#pragma once

#include <ceres/ceres.h>
#include <ceres/cost_function.h>
#include <opencv2/opencv.hpp>

#include "bundleadjust/Dataloader.h"
#include "BalDataloader.h"

class BAConstraint {
public:
    // observation = (x, y, unused); estimatedPose = column-major 4x4 (kept, the BAL model does not use it)
    BAConstraint(const Eigen::Vector3f& observation, const float * estimatedPose);

    // the ceres::AutoDiffCostFunction<BAConstraint, 2, 3, 3, 3, 6> the reference builds for every observation
    static ceres::CostFunction* create(const Eigen::Vector3f& observation, const float * estimatedPose);

    // parameter blocks: point[3] | rot[3] angle-axis | tr[3] | intrinsics[6] = fx fy ox oy k1 k2; residuals[2] = predicted - observed
    template <typename T>
    bool operator()(const T* const point, const T* const rot, const T* const tr, const T* const intrinsics, T* residuals) const;

    // prints the intermediate quantities of one evaluation (debugging aid of the reference)
    bool printOp(double * point, double * rot, double * tr, double * intrinsics, const float * estimatedPose) const;

    // Which algebra the functor evaluates: false (default) the 9-parameter BAL camera, true the reference's functor as it
    // is checked in (src/BAConstraint.cpp:15-80: [R | tr]^-1, fx fy ox oy, no distortion, observed - projected) — libmiba's
    // camera model #2.  BundleAdjustment sets it from BA_CAMERA_MODEL=reference.
    static void useReferenceModel(bool on);
    static bool referenceModel();

protected:
    const Eigen::Vector3f observation;
    Eigen::Matrix4f pose;
};
