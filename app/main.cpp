// Same call sequence as /root/reference/app/main.cpp:11-46 with the BalDataloader line enabled
// (the reference has it commented out because BalDataloader is abstract there).
// Usage: ba_main <bal-file> [solver: ITERATIVE_SCHUR|SPARSE_SCHUR|SPARSE_NORMAL_CHOLESKY] [max_iterations]
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "bundleadjust/BundleAdjustment.h"
#include "bundleadjust/BalDataloader.h"


int main(int argc, char **argv) {
    if (argc < 2) {
        std::cerr << "usage: " << argv[0] << " <problem-*-pre.txt> [solver] [max_iterations]" << std::endl;
        return 2;
    }

    BalDataloader* data = new BalDataloader(argv[1]);
    if (getenv("BA_SIGMA_CAM")) data->sigmaCam = atof(getenv("BA_SIGMA_CAM"));
    if (getenv("BA_SIGMA_X")) data->sigmaX = atof(getenv("BA_SIGMA_X"));

    ceres::LinearSolverType first = ceres::SPARSE_NORMAL_CHOLESKY;
    if (argc > 2) {
        if (!strcmp(argv[2], "ITERATIVE_SCHUR")) first = ceres::ITERATIVE_SCHUR;
        else if (!strcmp(argv[2], "SPARSE_SCHUR")) first = ceres::SPARSE_SCHUR;
        else if (!strcmp(argv[2], "DENSE_SCHUR")) first = ceres::DENSE_SCHUR;
    }

    auto solvers = {first};
    for (auto & solver : solvers) {
        ceres::Solver::Options options;

        options.trust_region_strategy_type = ceres::LEVENBERG_MARQUARDT;
        options.use_nonmonotonic_steps = false;
        options.linear_solver_type = solver;
        options.minimizer_progress_to_stdout = true;
        options.max_num_iterations = argc > 3 ? atoi(argv[3]) : 80;
        options.num_threads = 4;

        BundleAdjustment ba{data, options};

        ba.createProblem();
        ba.solve();
        ba.WriteToPLYFile("constrainTransReal.ply");
        std::cout << "FINAL_COST " << ba.lastSummary().final_cost << " ITERATIONS "
                  << ba.lastSummary().num_successful_steps + ba.lastSummary().num_unsuccessful_steps << std::endl;
    }

    delete data;

    return 0;
}
