#!/bin/bash
# in-run comparison of tile sizes (MIBA_TILE_CHUNKS) on the product and on the headline bench
for rep in 1 2; do
  for tc in "$@"; do
    echo -n "tc=$tc product "
    MIBA_TILE_CHUNKS=$tc timeout -k 10 120 python tools/time_k3.py final-13682 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(min(d['matvec_ms']), d['slots'], d['tiles'])" || exit 1
    echo -n "tc=$tc bench "
    MIBA_TILE_CHUNKS=$tc timeout -k 10 200 python bench.py --cpu-iters 0 --no-other-states --ttt 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k[:4]:v['avg_ms'] for k,v in d['kernels'].items()}, d['setup']['set_problem_s'])" || exit 1
  done
done
