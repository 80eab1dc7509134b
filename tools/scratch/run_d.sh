set -o pipefail
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_d -o d -- python3 -u $GRAFT_REPO_ROOT/tools/time_exact.py trafalgar-257 > $GRAFT_REPO_ROOT/gpurun_out/prof_d.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT && python tools/gap_analysis.py gpurun_out/prof_d/d_results.db > gpurun_out/gap_d.txt 2>&1
