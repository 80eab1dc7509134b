// The problem object the application drives: construct on a Dataloader, createProblem(), solve(), read the state back
// through the getters or as a PLY file.
//
// Same public members as the reference's class (/root/reference/include/bundleadjust/BundleAdjustment.h:13-39), so
// app/main.cpp and the mesh callback compile unchanged.  Behind them the ceres::Problem is replaced by a libmiba
// handle (include/miba.h): createProblem() hands the observations to miba_set_problem (filter, tiling, upload),
// solve() runs the Levenberg-Marquardt loop on the GPU, and the state arrays below are refreshed from the device
// whenever callbacks run and when solve() returns.
#pragma once

#include <string>

#include <ceres/ceres.h>

#include "Dataloader.h"

struct miba_handle_s;        // libmiba problem handle
struct PlyQueue;             // background PLY writer (BundleAdjustment.cpp)
class MeshWriterCallback;    // per-iteration dump, pushed onto options.callbacks by solve()

class BundleAdjustment {
public:
    // allocates the state arrays and lets the dataset fill them (Dataloader::initialize)
    BundleAdjustment(Dataloader* dataset, ceres::Solver::Options options);
    virtual ~BundleAdjustment();

    Dataloader* dataset;

    // --- the reference's call sequence (app/main.cpp:30-34) ---------------------------------------------------------
    void createProblem();
    void solve();
    void WriteToPLYFile(std::string filename);

    // --- current state, one block per camera / point (pointers into the arrays solve() keeps up to date) ---------------
    double* getPoint(size_t pointIndex);            // [3]
    double* getRotation(size_t cameraIndex);        // [3] angle-axis
    double* getTranslation(size_t cameraIndex);     // [3]
    double* getIntrinsics(size_t cameraIndex);      // [6] fx fy ox oy k1 k2

    // --- not in the reference ------------------------------------------------------------------------------------------
    // summary of the last solve() (the reference only prints it, src/BundleAdjustment.cpp:110)
    const ceres::Solver::Summary& lastSummary() const { return summary; }
    // the same PLY written by a background thread from a snapshot of the current state (used by the per-iteration mesh
    // callback; BA_ASYNC_PLY=0 makes it synchronous).  flushPlyWrites() waits for the queue; solve(), WriteToPLYFile()
    // and the destructor call it.
    void WriteToPLYFileAsync(std::string filename);
    void flushPlyWrites();

private:
    ceres::Solver::Options options;
    ceres::Solver::Summary summary;
    miba_handle_s* problem;      // device-resident problem (a ceres::Problem in the reference)
    MeshWriterCallback* callback;
    PlyQueue* plyQueue;

    // state: angle-axis rotations, translations, intrinsics (6 per camera), points
    double *R, *T, *intrinsics, *X;
};
