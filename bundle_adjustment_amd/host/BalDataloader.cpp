// BAL text loader, behaviour of /root/reference/src/BalDataloader.cpp:9-163:
// every scalar is parsed as float (:32-33,:44-45,:55-56) and widened to double only in
// initialize (:131-157), which perturbs the state with Marsaglia-polar normals drawn from the
// C library's rand() (unseeded => glibc seed-1 stream), six per camera then three per point.
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <tuple>

#include "bundleadjust/BalDataloader.h"

BalDataloader::BalDataloader(std::string path) {
    // <num_cameras> <num_points> <num_observations>
    // <camera_index> <point_index> <x> <y>        (num_observations lines)
    // 9 scalars per camera: R3, t3, f, k1, k2      (one per line)
    // 3 scalars per point
    std::ifstream file(path);
    if (!file.is_open()) { throw std::invalid_argument("File could not be opened"); }

    file >> num_camera >> num_points >> num_observations;
    if (!file || num_camera < 0 || num_points < 0 || num_observations < 0) {
        throw std::invalid_argument("Not a BAL file (bad header)");
    }

    obs_cam.resize(num_observations);
    obs_point.resize(num_observations);
    observations.resize(num_observations);
    for (int i = 0; i < num_observations; ++i) {
        int cam_index, point_index;
        float x, y;
        file >> cam_index >> point_index >> x >> y;
        observations[i] = std::pair<float, float>(x, y);
        obs_cam[i] = cam_index;
        obs_point[i] = point_index;
    }

    cameras.resize(num_camera);
    for (int i = 0; i < num_camera; ++i) {
        Camera &c = cameras[i];
        file >> c.R[0] >> c.R[1] >> c.R[2] >> c.t[0] >> c.t[1] >> c.t[2] >> c.f >> c.k1 >> c.k2;
    }

    points.resize(num_points);
    for (int i = 0; i < num_points; ++i) {
        float x, y, z;
        file >> x >> y >> z;
        points[i] = std::make_tuple(x, y, z);
    }
    if (!file) { throw std::invalid_argument("BAL file is truncated"); }
}

BalDataloader::~BalDataloader() {}

std::vector<cv::Point2f> BalDataloader::getObservations() const {
    std::vector<cv::Point2f> obs;
    obs.reserve(observations.size());
    for (auto &pt : observations) obs.emplace_back(pt.first, pt.second);
    return obs;
}

int BalDataloader::getObsCam(int index) const { return (int) obs_cam[index]; }
int BalDataloader::getObsPoint(int index) const { return (int) obs_point[index]; }
int BalDataloader::getNumPoints() const { return num_points; }
int BalDataloader::getNumObservations() const { return num_observations; }
int BalDataloader::getNumFrames() const { return num_camera; }
bool BalDataloader::isColorAvailable() const { return false; }
bool BalDataloader::isDepthAvailable() const { return false; }
cv::Mat BalDataloader::getColor(int) const { return cv::Mat(); }
cv::Mat BalDataloader::getDepth(int) const { return cv::Mat(); }

namespace {
// uniform in [0,1]
inline double RandDouble() { return static_cast<double>(rand()) / RAND_MAX; }

// Marsaglia polar method; the second variate is discarded as in the reference
inline double RandNormal() {
    double x1, x2, w;
    do {
        x1 = 2.0 * RandDouble() - 1.0;
        x2 = 2.0 * RandDouble() - 1.0;
        w = x1 * x1 + x2 * x2;
    } while (w >= 1.0 || w == 0.0);
    w = sqrt((-2.0 * log(w)) / w);
    return x1 * w;
}
}  // namespace

void BalDataloader::initialize(double *R, double *T, double *intrinsics, double *X) {
    for (int i = 0; i < num_camera; ++i) {
        const Camera &cam = cameras[i];
        for (int k = 0; k < 3; ++k) R[3 * i + k] = cam.R[k] + RandNormal() * sigmaCam;
        for (int k = 0; k < 3; ++k) T[3 * i + k] = cam.t[k] + RandNormal() * sigmaCam;
        intrinsics[6 * i + 0] = cam.f;  // fx
        intrinsics[6 * i + 1] = cam.f;  // fy
        intrinsics[6 * i + 2] = 0;      // ox
        intrinsics[6 * i + 3] = 0;      // oy
        intrinsics[6 * i + 4] = cam.k1;
        intrinsics[6 * i + 5] = cam.k2;
    }
    for (int i = 0; i < num_points; ++i) {
        const auto &pt = points[i];
        X[3 * i + 0] = std::get<0>(pt) + RandNormal() * sigmaX;
        X[3 * i + 1] = std::get<1>(pt) + RandNormal() * sigmaX;
        X[3 * i + 2] = std::get<2>(pt) + RandNormal() * sigmaX;
    }
}

Eigen::Vector3i BalDataloader::getPointColor(int) const {
    Eigen::Vector3i color;
    color.setConstant(255);
    return color;
}

const float *BalDataloader::getEstimatedPose(int) const {
    static const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    return identity;
}
