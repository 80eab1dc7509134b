#!/bin/bash
# A/B of the product S*v inside ONE gpurun call (box-to-box variation is 3-5 %).
# usage: tools/ab_k3_r3.sh variant...   where a variant is  name[,ENV=value...][,lib=path]
cfg=${AB_CFG:-final-13682}
for rep in 1 2 3; do
  for v in "$@"; do
    IFS=, read -ra parts <<< "$v"
    envs=()
    for p in "${parts[@]:1}"; do
      case $p in lib=*) envs+=("MIBA_LIB=${p#lib=}");; *) envs+=("$p");; esac
    done
    echo -n "${parts[0]} "
    env "${envs[@]}" timeout -k 10 150 python tools/time_k3.py $cfg 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(min(d['matvec_ms']), d['Sv_norm'])" || exit 1
  done
done
