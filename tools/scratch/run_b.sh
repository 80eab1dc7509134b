set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u bench.py > gpurun_out/r2_bench_b.json 2> gpurun_out/r2_bench_b.err || exit 1
timeout -k 10 300 python -u tools/time_exact.py > gpurun_out/r2_small_configs.jsonl 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b -o b -- python3 -u $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-iters 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_b.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT && python tools/gap_analysis.py gpurun_out/prof_b/b_results.db > gpurun_out/gap_b.txt 2>&1
