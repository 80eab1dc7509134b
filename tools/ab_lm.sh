#!/bin/bash
# In-run A/B of the LM iteration (driver's 20 steps at Final-13682): variants are  name[,ENV=value...][,lib=path]
R=${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
  for v in "$@"; do
    IFS=, read -ra parts <<< "$v"
    envs=()
    for p in "${parts[@]:1}"; do
      case $p in lib=*) envs+=("MIBA_LIB=${p#lib=}");; *) envs+=("$p");; esac
    done
    echo -n "${parts[0]} "
    env "${envs[@]}" timeout -k 10 200 python3 -u $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-states --no-exact --ttt 0 > /tmp/ab.json 2> /tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
    python3 -c "import json; d=json.load(open('/tmp/ab.json')); k=d['kernels']; print('LM it/s %.2f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'K2a %.4f' % k['K2a_mf_init_points']['avg_ms'], 'K2b %.4f' % k['K2b_mf_camrows']['avg_ms'], 'K4 %.4f' % k['K4_mf_backsub']['avg_ms'], 'K3 %.4f' % k['K3_mf_matvec']['avg_ms'])"
  done
done
