#!/bin/bash
# GPU test suite under every environment knob of libmiba (run on the GPU box: bash tools/gpu_env_matrix.sh)
set -o pipefail
run() {
  echo "== $*"
  env "$@" timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread $EXTRA > /tmp/env_matrix_run.log 2>&1
  local rc=$?
  if [ $rc -ne 0 ]; then tail -60 /tmp/env_matrix_run.log; return 1; fi
  tail -1 /tmp/env_matrix_run.log
}
EXTRA=""
if [ -z "$SKIP_TILE_CHUNKS" ]; then
run MIBA_TILE_CHUNKS=1 || exit 1
run MIBA_TILE_CHUNKS=16 || exit 1
fi
if [ -z "$SKIP_DONE" ]; then
run MIBA_POINT_ORDER=0 || exit 1
fi
if [ -z "$SKIP_DONE" ]; then
run MIBA_FORCE_RCCL=1 || exit 1
fi
run MIBA_DENSE_PLAN=0 || exit 1
run MIBA_DENSE_PLAN=host || exit 1
run MIBA_DEVICE_PLAN=0 || exit 1
run MIBA_K3=pm || exit 1
run MIBA_K4=pm || exit 1
run MIBA_K1_FOLD=0 || exit 1
run MIBA_TAIL_SPLIT=1 || exit 1
run MIBA_SPARSE_ARCS=1 || exit 1
run MIBA_SPARSE_ARCS=2 || exit 1
run MIBA_SPARSE_ARCS=16 || exit 1
run MIBA_SPARSE_NESTED=0 || exit 1
run MIBA_SPARSE_AHEAD=0 || exit 1
run MIBA_XCD_REMAP=0 || exit 1
run MIBA_ALLREDUCE=peer MIBA_TAIL_NO_FUSE=1 || exit 1
# the tests of the exact solve and the multi-rank runs that use it are left out when it is switched off
EXTRA="--deselect tests/test_gpu_dense.py --deselect tests/test_gpu_sparse.py --deselect tests/test_gpu_multirank.py"
run MIBA_DENSE=0 || exit 1
