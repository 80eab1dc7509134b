#!/bin/bash
# exact solve at Final-13682 with the number of arcs chosen by the symbolic phase (auto) and fixed (on the GPU box): ms per linear solve from bench.py's exact_solve block
R=${GRAFT_REPO_ROOT:-.}
for arcs in ${ARCS:-auto 1 3 8 16}; do
  if [ $arcs = auto ]; then unset MIBA_SPARSE_ARCS; else export MIBA_SPARSE_ARCS=$arcs; fi
  MIBA_SPARSE_DEBUG=1 timeout -k 10 200 python3 -u $R/bench.py --steps 2 --warmup 1 --cpu-iters 0 --no-other-states --ttt 0 > /tmp/ex.json 2> /tmp/ex.err || { tail -5 /tmp/ex.err; exit 1; }
  grep "tile order with\|tile factorisation" /tmp/ex.err | sort -u
  python3 -c "import json; d=json.load(open('/tmp/ex.json'))['exact_solve']; print('arcs', '$arcs', {k: d[k] for k in ('ms_per_lm_iteration','ms_per_linear_solve','first_solve_s')})"
done
