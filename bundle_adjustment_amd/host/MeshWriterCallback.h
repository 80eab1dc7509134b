// Per-iteration PLY dump, as /root/reference/src/MeshWriterCallback.h:11-28.  Disabled when the
// environment variable BA_ITERATION_PLY is "0" (a Final-13682 dump is ~150 MB of text per iteration).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <string>

#include <ceres/ceres.h>

#include <bundleadjust/BundleAdjustment.h>

class MeshWriterCallback : public ceres::IterationCallback {
public:
    explicit MeshWriterCallback(BundleAdjustment* ba, std::string prefix):ba(ba), prefix(prefix){
        const char* e = getenv("BA_ITERATION_PLY");
        enabled = !(e && e[0] == '0');
    }

    ~MeshWriterCallback() {}

    // false: BA_ITERATION_PLY=0, the callback would do nothing (solve() then leaves it out, so that no state is
    // copied back per iteration on its account)
    bool isEnabled() const { return enabled; }

    ceres::CallbackReturnType operator()(const ceres::IterationSummary &summary) {
        if(enabled && summary.iteration % 1 == 0) {
            char name[256];
            snprintf(name, sizeof(name), "mesh_%s_%d.ply", prefix.c_str(), summary.iteration);
            ba->WriteToPLYFileAsync(name);   // snapshot now, text later: the LM loop does not wait for the file
        }
        return ceres::SOLVER_CONTINUE;
    }

private:
    BundleAdjustment* ba;
    std::string prefix;
    bool enabled;
};
