// Problem API, same surface as /root/reference/include/bundleadjust/BundleAdjustment.h:13-39.
// ceres::Problem / ceres::Solve are replaced by a libmiba handle (include/miba.h).
#pragma once

#include <string>

#include <ceres/ceres.h>

#include "Dataloader.h"

class MeshWriterCallback;
struct miba_handle_s;
struct PlyQueue;

class BundleAdjustment {
public:
    BundleAdjustment(Dataloader* dataset, ceres::Solver::Options options);
    virtual ~BundleAdjustment();

    void createProblem();
    void solve();

    double* getRotation(size_t cameraIndex);
    double* getTranslation(size_t cameraIndex);
    double* getIntrinsics(size_t cameraIndex);
    double* getPoint(size_t pointIndex);

    void WriteToPLYFile(std::string filename);
    // Not in the reference: the same file written by a background thread from a snapshot of the current state (used by
    // the per-iteration mesh callback; BA_ASYNC_PLY=0 makes it synchronous).  flushPlyWrites() waits for the queue;
    // solve(), WriteToPLYFile() and the destructor call it.
    void WriteToPLYFileAsync(std::string filename);
    void flushPlyWrites();

    Dataloader* dataset;

    // result of the last solve() (the reference only prints it, src/BundleAdjustment.cpp:110)
    const ceres::Solver::Summary& lastSummary() const { return summary; }

private:
    ceres::Solver::Options options;
    miba_handle_s* problem;   // device-resident problem (was ceres::Problem)
    ceres::Solver::Summary summary;

    double *R;
    double *X;
    double *T;
    double *intrinsics;
    MeshWriterCallback *callback;
    PlyQueue *plyQueue;
};
