#!/bin/bash
# A/B of whole LM iterations inside ONE gpurun call: tools/ab_bench_r3.sh variant...  (variant = name[,ENV=value...][,lib=path])
for rep in 1 2; do
  for v in "$@"; do
    IFS=, read -ra parts <<< "$v"
    envs=()
    for p in "${parts[@]:1}"; do
      case $p in lib=*) envs+=("MIBA_LIB=${p#lib=}");; *) envs+=("$p");; esac
    done
    echo -n "${parts[0]} "
    env "${envs[@]}" timeout -k 10 200 python bench.py --steps ${AB_STEPS:-20} --warmup ${AB_WARMUP:-5} --no-cpu-baseline --no-other-states --no-exact --ttt 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d.get('kernels',{})
print(d['value'], d['ms_per_step'], 'pcg_ms', d.get('ms_per_pcg_iteration'), {n:round(v.get('avg_launch_ms',0),4) for n,v in k.items()} if isinstance(k,dict) else '', 'K3', d['roofline'].get('avg_launch_ms'))" || exit 1
  done
done
