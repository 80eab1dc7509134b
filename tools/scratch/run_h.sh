cd $GRAFT_REPO_ROOT
MIBA_TAIL_DEBUG=1 timeout -k 10 400 python -u bench.py --cpu-iters 0 --steps 3 --warmup 1 > gpurun_out/dbg.json 2> gpurun_out/dbg.err || exit 1
MIBA_TAIL_DEBUG=1 timeout -k 10 200 python -u tools/time_exact.py trafalgar-257 > gpurun_out/dbg2.json 2> gpurun_out/dbg2.err
