// Prints residual and autodiff Jacobian of BAConstraint::create(...) for records read from stdin
// ("cam9 X3 obs2" per line, 14 hex/decimal doubles): the host-side functor checked against the
// mpmath fixtures by tests/test_host_cpp.py.  Needs no GPU.
#include <chrono>
#include <fstream>
#include <cstdio>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "bundleadjust/BAConstraint.h"
#include "bundleadjust/BalDataloader.h"
#include "bundleadjust/BundleAdjustment.h"

// "--loader <file>": dump what BalDataloader parsed and what initialize() produced
static int dump_loader(const char *path) {
    BalDataloader d(path);
    printf("%d %d %d\n", d.getNumFrames(), d.getNumPoints(), d.getNumObservations());
    std::vector<cv::Point2f> obs = d.getObservations();
    for (int i = 0; i < d.getNumObservations(); ++i)
        printf("%d %d %a %a\n", d.getObsCam(i), d.getObsPoint(i), (double) obs[i].x, (double) obs[i].y);
    std::vector<double> R(3 * d.getNumFrames()), T(3 * d.getNumFrames()), K(6 * d.getNumFrames()), X(3 * d.getNumPoints());
    d.initialize(R.data(), T.data(), K.data(), X.data());
    for (double x : R) printf("%a\n", x);
    for (double x : T) printf("%a\n", x);
    for (double x : K) printf("%a\n", x);
    for (double x : X) printf("%a\n", x);
    Eigen::Vector3i c = d.getPointColor(0);
    printf("%d %d %d %d %d %d\n", c(0), c(1), c(2), (int) d.isColorAvailable(), (int) d.isDepthAvailable(),
           (int) (d.getEstimatedPose(0)[0] == 1.f && d.getEstimatedPose(0)[5] == 1.f && d.getEstimatedPose(0)[1] == 0.f));
    return 0;
}

// "--loader-time <file>": seconds spent in the BalDataloader constructor, and the same file through the
// reference's serial `ifstream >>` loop (src/BalDataloader.cpp:21-58) for comparison
static int time_loader(const char *path) {
    auto t0 = std::chrono::steady_clock::now();
    BalDataloader d(path);
    const double fast = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    t0 = std::chrono::steady_clock::now();
    std::ifstream file(path);
    int nc, np, no;
    file >> nc >> np >> no;
    double chk = 0.0;
    for (int i = 0; i < no; ++i) { int c, p; float x, y; file >> c >> p >> x >> y; chk += x; }
    for (long long i = 0; i < 9LL * nc + 3LL * np; ++i) { float v; file >> v; chk += v; }
    const double serial = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("cameras %d points %d observations %d fast_s %.3f serial_ifstream_s %.3f speedup %.1f checksum %g\n", d.getNumFrames(),
           d.getNumPoints(), d.getNumObservations(), fast, serial, serial / fast, chk);
    return 0;
}

// "--ply <file> <out.ply>": BundleAdjustment constructed on the loaded data (no createProblem, no GPU), its state dumped
// with WriteToPLYFile; the state itself goes to stdout as hex doubles (translations, then points) so that the caller
// can format the expected text independently.  The sequence of the reference's tests/procrustes/procrustes_test.cpp:27-34.
static int dump_ply(const char *path, const char *out) {
    BalDataloader d(path);
    ceres::Solver::Options options;
    BundleAdjustment ba{&d, options};
    ba.WriteToPLYFile(out);
    printf("%d %d\n", d.getNumFrames(), d.getNumPoints());
    for (int i = 0; i < d.getNumFrames(); ++i) printf("%a %a %a\n", ba.getTranslation(i)[0], ba.getTranslation(i)[1], ba.getTranslation(i)[2]);
    for (int i = 0; i < d.getNumPoints(); ++i) printf("%a %a %a\n", ba.getPoint(i)[0], ba.getPoint(i)[1], ba.getPoint(i)[2]);
    return 0;
}

int main(int argc, char **argv) {
    if (argc == 4 && std::string(argv[1]) == "--ply") {
        try { return dump_ply(argv[2], argv[3]); } catch (const std::exception &e) { printf("exception %s\n", e.what()); return 3; }
    }
    if (argc == 3 && std::string(argv[1]) == "--loader-time") {
        try { return time_loader(argv[2]); } catch (const std::invalid_argument &e) { printf("invalid_argument %s\n", e.what()); return 3; }
    }
    if (argc == 3 && std::string(argv[1]) == "--loader") {
        try { return dump_loader(argv[2]); } catch (const std::invalid_argument &e) { printf("invalid_argument %s\n", e.what()); return 3; }
    }
    double v[15];
    static const float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (argc == 2 && std::string(argv[1]) == "--model2") {
        // the reference's live functor: records rot[3] tr[3] fx fy ox oy X[3] obs[2]; prints r, d/drot, d/dtr (2 x 3 each), d/dX
        BAConstraint::useReferenceModel(true);
        while (true) {
            for (int i = 0; i < 15; ++i) if (scanf("%la", &v[i]) != 1) return 0;
            Eigen::Vector3f obs;
            obs(0) = (float) v[13]; obs(1) = (float) v[14]; obs(2) = 1.f;
            std::unique_ptr<ceres::CostFunction> cf(BAConstraint::create(obs, id));
            double point[3] = {v[10], v[11], v[12]}, rot[3] = {v[0], v[1], v[2]}, tr[3] = {v[3], v[4], v[5]};
            double K[6] = {v[6], v[7], v[8], v[9], 0.1, -0.2};   // k1, k2 must not matter
            const double *params[4] = {point, rot, tr, K};
            double r[2], Jp[6], Jr[6], Jt[6], Jk[12];
            double *jac[4] = {Jp, Jr, Jt, Jk};
            if (!cf->Evaluate(params, r, jac)) return 1;
            printf("%a %a", r[0], r[1]);
            for (int i = 0; i < 6; ++i) printf(" %a", Jr[i]);
            for (int i = 0; i < 6; ++i) printf(" %a", Jt[i]);
            for (int i = 0; i < 6; ++i) printf(" %a", Jp[i]);
            printf(" %a %a\n", Jk[4] + Jk[5], Jk[10] + Jk[11]);   // d/dk1 + d/dk2: zero
        }
    }
    while (true) {
        for (int i = 0; i < 14; ++i) if (scanf("%la", &v[i]) != 1) return 0;
        Eigen::Vector3f obs;
        obs(0) = (float) v[12]; obs(1) = (float) v[13]; obs(2) = 1.f;
        std::unique_ptr<ceres::CostFunction> cf(BAConstraint::create(obs, id));
        double point[3] = {v[9], v[10], v[11]}, rot[3] = {v[0], v[1], v[2]}, tr[3] = {v[3], v[4], v[5]};
        double K[6] = {v[6], v[6], 0, 0, v[7], v[8]};
        const double *params[4] = {point, rot, tr, K};
        double r[2], Jp[6], Jr[6], Jt[6], Jk[12];
        double *jac[4] = {Jp, Jr, Jt, Jk};
        if (!cf->Evaluate(params, r, jac)) return 1;
        printf("%a %a", r[0], r[1]);
        for (int i = 0; i < 2; ++i) {   // camera block in libmiba order: w t f k1 k2 (f = dfx + dfy)
            for (int j = 0; j < 3; ++j) printf(" %a", Jr[i * 3 + j]);
            for (int j = 0; j < 3; ++j) printf(" %a", Jt[i * 3 + j]);
            printf(" %a %a %a", Jk[i * 6 + 0] + Jk[i * 6 + 1], Jk[i * 6 + 4], Jk[i * 6 + 5]);
        }
        for (int i = 0; i < 6; ++i) printf(" %a", Jp[i]);
        printf("\n");
    }
}
