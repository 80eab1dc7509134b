set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_dense.py tests/test_gpu_sparse.py tests/test_zz_no_alloc.py -m gpu -x -q -s --timeout 600 --timeout-method thread > gpurun_out/sp.log 2>&1 || exit 1
timeout -k 10 300 python -u tools/time_exact.py > gpurun_out/r2_small_configs.jsonl 2>&1 || exit 1
