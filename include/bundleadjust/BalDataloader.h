// Bundle Adjustment in the Large text files (https://grail.cs.washington.edu/projects/bal/) as a Dataloader.
//
// Public surface of the reference's loader (/root/reference/include/bundleadjust/BalDataloader.h:13-71): the Camera
// record, the public containers the parse fills, and the Dataloader overrides.  The parse itself is multi-threaded and
// can be cached (bundle_adjustment_amd/host/BalDataloader.cpp); every scalar is still rounded through float exactly as
// the reference's `ifstream >> float` does.
#pragma once

#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "bundleadjust/Dataloader.h"

// one camera line block of the file: angle-axis rotation, translation, focal length, two radial distortion terms
struct Camera {
    float R[3];
    float t[3];
    float f, k1, k2;
};

class BalDataloader : public Dataloader {
public:
    BalDataloader(std::string path);   // throws std::invalid_argument on a missing or malformed file
    ~BalDataloader() override;

    // ---- what the file holds, in file order -------------------------------------------------------------------------
    int num_camera, num_points, num_observations;
    std::vector<size_t> obs_cam;                          // observation i was made by camera obs_cam[i]
    std::vector<size_t> obs_point;                        // ... of point obs_point[i]
    std::vector<std::pair<float, float>> observations;    // ... at this pixel
    std::vector<Camera> cameras;
    std::vector<std::tuple<float, float, float>> points;

    // ---- noise of initialize(): the reference hard-codes these two values (src/BalDataloader.cpp:132-133) -------------
    double sigmaCam = 0.01;   // on rotation and translation
    double sigmaX = 2;        // on points

    // ---- Dataloader -----------------------------------------------------------------------------------------------------
    int getNumFrames() const override;
    int getNumPoints() const override;
    int getNumObservations() const override;
    std::vector<cv::Point2f> getObservations() const override;
    int getObsCam(int index) const override;
    int getObsPoint(int index) const override;
    void initialize(double* R, double* T, double* intrinsics, double* X) override;
    // pure virtual in the reference (its BalDataloader is abstract); BAL has no pose prior: identity
    const float* getEstimatedPose(int index) const override;

    Eigen::Vector3i getPointColor(int point_index) const override;   // BAL has no colours: a constant
    bool isColorAvailable() const override;
    bool isDepthAvailable() const override;
    cv::Mat getColor(int frameId) const override;
    cv::Mat getDepth(int frameId) const override;
};
