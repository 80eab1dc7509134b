#!/bin/bash
# A/B of libmiba builds on the headline bench inside ONE gpurun call: tools/ab_bench.sh lib1 lib2 ...
for rep in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib "
    MIBA_LIB=$lib timeout -k 10 200 python bench.py --cpu-iters 0 --no-other-states --ttt 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k[:4]: v['avg_ms'] for k, v in d['kernels'].items() if v['launches']})" || exit 1
  done
done
