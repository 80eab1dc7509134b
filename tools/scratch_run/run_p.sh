set -o pipefail
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_sparse -o sp -- python3 -u $GRAFT_REPO_ROOT/tools/scratch_run/sparse_prof.py > $GRAFT_REPO_ROOT/gpurun_out/sparse3.log 2>&1
