#!/bin/bash
# In-run A/B of the exact solve (bench.py's exact_solve block at Final-13682): variants are  name[,ENV=value...][,lib=path]
R=${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
  for v in "$@"; do
    IFS=, read -ra parts <<< "$v"
    envs=()
    for p in "${parts[@]:1}"; do
      case $p in lib=*) envs+=("MIBA_LIB=${p#lib=}");; *) envs+=("$p");; esac
    done
    echo -n "${parts[0]} "
    env "${envs[@]}" timeout -k 10 200 python3 -u $R/bench.py --steps 2 --warmup 1 --cpu-iters 0 --no-other-states --ttt 0 > /tmp/ex.json 2> /tmp/ex.err || { tail -5 /tmp/ex.err; exit 1; }
    python3 -c "import json; d=json.load(open('/tmp/ex.json'))['exact_solve']; print(d['ms_per_linear_solve'], d['ms_per_lm_iteration'])"
  done
done
