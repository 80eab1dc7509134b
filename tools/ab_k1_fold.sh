#!/bin/bash
# In-run A/B (one box, alternating): the candidate's cost inside the back-substitution (default) against its own kernel (MIBA_K1_FOLD=0).
R=${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
  for fold in 1 0; do
    MIBA_K1_FOLD=$fold timeout -k 10 200 python3 -u $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-states --no-exact --ttt 0 > /tmp/ab.json 2> /tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
    python3 -c "import json; d=json.load(open('/tmp/ab.json')); k=d['kernels']; print('fold', $fold, 'LM it/s %.2f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'K4 %.4f' % k['K4_mf_backsub']['avg_ms'], 'K1p %.4f x %d' % (k['K1p_mf_cost']['avg_ms'], k['K1p_mf_cost']['launches']))"
  done
done
