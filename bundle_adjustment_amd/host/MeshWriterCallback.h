// Iteration callback that dumps the current state as mesh_<prefix>_<iteration>.ply, the job of the reference's
// callback of the same name (/root/reference/src/MeshWriterCallback.h:11-28).  Here the dump is a snapshot handed
// to BundleAdjustment's background writer, so the Levenberg-Marquardt loop does not wait for the file; the
// environment variable BA_ITERATION_PLY=0 switches the dumps off (a Final-13682 dump is ~150 MB of text).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <string>

#include <ceres/ceres.h>

#include <bundleadjust/BundleAdjustment.h>

class MeshWriterCallback : public ceres::IterationCallback {
public:
    explicit MeshWriterCallback(BundleAdjustment* owner, std::string file_prefix) : ba(owner), prefix(std::move(file_prefix)) {
        const char* knob = getenv("BA_ITERATION_PLY");
        enabled = !(knob && knob[0] == '0');
    }

    // false: the callback would do nothing; solve() then leaves it out of options.callbacks, so that no state is copied
    // back from the device per iteration on its account
    bool isEnabled() const { return enabled; }

    ceres::CallbackReturnType operator()(const ceres::IterationSummary& summary) {
        if (!enabled) return ceres::SOLVER_CONTINUE;
        char name[256];
        snprintf(name, sizeof(name), "mesh_%s_%d.ply", prefix.c_str(), summary.iteration);
        ba->WriteToPLYFileAsync(name);
        return ceres::SOLVER_CONTINUE;
    }

private:
    BundleAdjustment* ba;
    std::string prefix;
    bool enabled;
};
