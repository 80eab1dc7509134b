// Data source of a bundle-adjustment problem: who saw what where, and a first guess of the state.
//
// The virtual functions are the ones the reference declares (/root/reference/include/bundleadjust/Dataloader.h:9-27),
// with the same names, signatures and constness, so that a loader written against the reference compiles against
// this header.  (The reference marks its pure virtuals `inline`, which is an ODR hazard; that keyword is not kept.)
// They are grouped here by what the solver side does with them.
#pragma once
#include <vector>

#include <Eigen/Dense>
#include <opencv2/opencv.hpp>

class Dataloader {
public:
    virtual ~Dataloader() = 0;

    // ---- sizes --------------------------------------------------------------------------------------------------
    virtual int getNumFrames() const = 0;          // cameras
    virtual int getNumPoints() const = 0;          // 3-D points
    virtual int getNumObservations() const = 0;    // 2-D measurements

    // ---- measurements: observation `index` is pixel getObservations()[index] of point getObsPoint(index)
    //      (-1: no point, the solver skips it) seen by camera getObsCam(index) ---------------------------------------
    virtual std::vector<cv::Point2f> getObservations() const = 0;
    virtual int getObsCam(int index) const = 0;
    virtual int getObsPoint(int index) const = 0;

    // ---- initial state, written into the caller's arrays: R[3 * frames] angle-axis, T[3 * frames],
    //      intrinsics[6 * frames] = fx fy ox oy k1 k2, X[3 * points] ----------------------------------------------------
    virtual void initialize(double* R, double* T, double* intrinsics, double* X) = 0;
    virtual const float* getEstimatedPose(int index) const = 0;   // column-major 4x4 prior of frame `index`

    // ---- appearance (PLY colours, the RGB-D front end) -----------------------------------------------------------
    virtual Eigen::Vector3i getPointColor(int index) const = 0;
    virtual bool isColorAvailable() const = 0;
    virtual bool isDepthAvailable() const = 0;
    virtual cv::Mat getColor(int frameId) const = 0;
    virtual cv::Mat getDepth(int frameId) const = 0;
};

inline Dataloader::~Dataloader() {}
