set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_model2.py tests/test_gpu_loss.py tests/test_gpu_multirank.py tests/test_gpu_edges.py -m gpu -x -q --timeout 600 --timeout-method thread -k "not 50k" > gpurun_out/tail2.log 2>&1 || exit 1
timeout -k 10 400 python -u bench.py --cpu-iters 0 > gpurun_out/r2_bench_c.json 2> gpurun_out/r2_bench_c.err || exit 1
timeout -k 10 300 python -u tools/time_exact.py > gpurun_out/r2_small_configs.jsonl 2>&1 || exit 1
MIBA_TAIL_DEBUG=1 timeout -k 10 400 python -u bench.py --cpu-iters 0 --steps 3 --warmup 1 > gpurun_out/dbg.json 2> gpurun_out/dbg.err || exit 1
MIBA_TAIL_DEBUG=1 timeout -k 10 200 python -u tools/time_exact.py trafalgar-257 > gpurun_out/dbg2.json 2> gpurun_out/dbg2.err
