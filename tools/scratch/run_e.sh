cd $GRAFT_REPO_ROOT
for c in 16 8 4 2 1; do echo "chunks $c"; MIBA_TILE_CHUNKS=$c timeout -k 10 200 python -u tools/time_exact.py ladybug-49 trafalgar-257 dubrovnik-356 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); it=d['iterative']
    print('  ', d['config'], 'us/PCG it %.1f' % (1e6*it['linear_solver_s']/it['linear_iterations']), 'ms/LM %.2f' % it['ms_per_lm_iteration'], 'lin its', it['linear_iterations'])
" || exit 1; done
