// Restatement of the two ceres/rotation.h templates the reference calls
// (ceres::AngleAxisToRotationMatrix at /root/reference/src/BAConstraint.cpp:35,148 and
// ceres::AngleAxisRotatePoint at :50,106), with Ceres' published branch rule:
// theta^2 > DBL_EPSILON -> Rodrigues, otherwise the first-order expansion.
#pragma once
#include <cmath>
#include <limits>

#include "ceres.h"

namespace ceres {

template <typename T>
inline void AngleAxisRotatePoint(const T angle_axis[3], const T pt[3], T result[3]) {
  using std::sqrt; using std::sin; using std::cos;
  const T theta2 = angle_axis[0] * angle_axis[0] + angle_axis[1] * angle_axis[1] + angle_axis[2] * angle_axis[2];
  if (theta2 > T(std::numeric_limits<double>::epsilon())) {
    const T theta = sqrt(theta2);
    const T costheta = cos(theta);
    const T sintheta = sin(theta);
    const T theta_inverse = T(1.0) / theta;
    const T w[3] = {angle_axis[0] * theta_inverse, angle_axis[1] * theta_inverse, angle_axis[2] * theta_inverse};
    const T w_cross_pt[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - costheta);
    result[0] = pt[0] * costheta + w_cross_pt[0] * sintheta + w[0] * tmp;
    result[1] = pt[1] * costheta + w_cross_pt[1] * sintheta + w[1] * tmp;
    result[2] = pt[2] * costheta + w_cross_pt[2] * sintheta + w[2] * tmp;
  } else {
    const T w_cross_pt[3] = {angle_axis[1] * pt[2] - angle_axis[2] * pt[1], angle_axis[2] * pt[0] - angle_axis[0] * pt[2],
                             angle_axis[0] * pt[1] - angle_axis[1] * pt[0]};
    result[0] = pt[0] + w_cross_pt[0];
    result[1] = pt[1] + w_cross_pt[1];
    result[2] = pt[2] + w_cross_pt[2];
  }
}

// column-major 3x3 output, as Ceres writes it
template <typename T>
inline void AngleAxisToRotationMatrix(const T* angle_axis, T* R) {
  using std::sqrt; using std::sin; using std::cos;
  const T theta2 = angle_axis[0] * angle_axis[0] + angle_axis[1] * angle_axis[1] + angle_axis[2] * angle_axis[2];
  if (theta2 > T(std::numeric_limits<double>::epsilon())) {
    const T theta = sqrt(theta2);
    const T wx = angle_axis[0] / theta, wy = angle_axis[1] / theta, wz = angle_axis[2] / theta;
    const T costheta = cos(theta), sintheta = sin(theta);
    R[0] = costheta + wx * wx * (T(1.0) - costheta);
    R[1] = wz * sintheta + wx * wy * (T(1.0) - costheta);
    R[2] = -wy * sintheta + wx * wz * (T(1.0) - costheta);
    R[3] = wx * wy * (T(1.0) - costheta) - wz * sintheta;
    R[4] = costheta + wy * wy * (T(1.0) - costheta);
    R[5] = wx * sintheta + wy * wz * (T(1.0) - costheta);
    R[6] = wy * sintheta + wx * wz * (T(1.0) - costheta);
    R[7] = -wx * sintheta + wy * wz * (T(1.0) - costheta);
    R[8] = costheta + wz * wz * (T(1.0) - costheta);
  } else {
    R[0] = T(1.0); R[1] = angle_axis[2]; R[2] = -angle_axis[1];
    R[3] = -angle_axis[2]; R[4] = T(1.0); R[5] = angle_axis[0];
    R[6] = angle_axis[1]; R[7] = -angle_axis[0]; R[8] = T(1.0);
  }
}

}  // namespace ceres
