for tc in 1 2 4 8; do
  echo "tc=$tc"
  MIBA_TILE_CHUNKS=$tc timeout -k 10 200 python tools/time_exact.py 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(' ', d['config'], 'iterative', round(d['iterative']['ms_per_lm_iteration'],3), d['iterative']['linear_iterations'], 'exact', round(d['exact']['ms_per_lm_iteration'],3))" || exit 1
done
