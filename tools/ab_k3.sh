#!/bin/bash
# A/B of libmiba builds on the product S*v inside ONE gpurun call (box-to-box variation is 3-5 %): tools/ab_k3.sh lib1 lib2 ...
for rep in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib "
    MIBA_LIB=$lib timeout -k 10 120 python tools/time_k3.py final-13682 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(min(d['matvec_ms']))" || exit 1
  done
done
