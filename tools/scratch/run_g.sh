set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 900 --timeout-method thread > gpurun_out/full.log 2>&1 || exit 1
timeout -k 10 400 python -u bench.py > gpurun_out/r2_bench_d.json 2> gpurun_out/r2_bench_d.err || exit 1
timeout -k 10 300 python -u tools/time_exact.py > gpurun_out/r2_small_configs.jsonl 2>&1 || exit 1
