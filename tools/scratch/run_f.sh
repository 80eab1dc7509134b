cd $GRAFT_REPO_ROOT
for c in 1 2 4 8; do echo "check_every $c"; CHECK_EVERY=$c timeout -k 10 200 python -u tools/time_exact.py ladybug-49 trafalgar-257 dubrovnik-356 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); it=d['iterative']
    print('  ', d['config'], 'us/PCG it %.1f' % (1e6*it['linear_solver_s']/it['linear_iterations']), 'ms/LM %.2f' % it['ms_per_lm_iteration'], 'lin its', it['linear_iterations'])
" || exit 1
timeout -k 10 300 python -u bench.py --cpu-iters 0 --check-every $c | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   final: value %.2f ms/step %.3f ms/pcg %.4f' % (d['value'], d['ms_per_step'], d['ms_per_pcg_iteration']))" || exit 1
done
