// Loader for BAL text files (https://grail.cs.washington.edu/projects/bal/), same public
// surface as /root/reference/include/bundleadjust/BalDataloader.h:13-71.
#pragma once

#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "bundleadjust/Dataloader.h"

struct Camera {
    float R[3];
    float t[3];
    float f, k1, k2;
};

class BalDataloader : public Dataloader {
public:
    BalDataloader(std::string path);
    ~BalDataloader() override;

    int num_camera, num_points, num_observations;

    std::vector<std::pair<float, float>> observations; // 2d points
    std::vector<size_t> obs_cam;   //  ith 2d point on jth camera
    std::vector<size_t> obs_point; //  ith 2d point corresponds to jth 3d point
    std::vector<Camera> cameras;
    std::vector<std::tuple<float, float, float>> points;

    // noise of initialize(); the reference hard-codes these values (src/BalDataloader.cpp:132-133)
    double sigmaCam = 0.01;
    double sigmaX = 2;

    int getObsCam(int index) const override;
    int getObsPoint(int index) const override;
    int getNumPoints() const override;
    std::vector<cv::Point2f> getObservations() const override;
    int getNumObservations() const override;
    int getNumFrames() const override;

    Eigen::Vector3i getPointColor(int point_index) const override;

    bool isColorAvailable() const override;
    bool isDepthAvailable() const override;

    cv::Mat getColor(int frameId) const override;
    cv::Mat getDepth(int frameId) const override;

    void initialize(double* R, double* T, double* intrinsics, double* X) override;

    // the reference leaves this pure virtual (BalDataloader is abstract there);
    // BAL has no pose prior: identity, column-major 4x4
    const float* getEstimatedPose(int index) const override;
};
