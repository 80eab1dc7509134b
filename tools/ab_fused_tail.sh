#!/bin/bash
# Rehearsal on ONE GPU (ranks of one process share device 0; absolute numbers mean little, the differences are the fixed cost per
# PCG iteration): the exchange of q inside the tail kernel against phase A + all-reduce kernel + rest of the tail, and against the
# host transport.  usage (on the GPU box): bash tools/ab_fused_tail.sh [ranks]   -> gpurun_out/ab_fused_tail.jsonl
set -o pipefail
R=${GRAFT_REPO_ROOT:-.}
N=${1:-2}
O=$R/gpurun_out/ab_fused_tail.jsonl
: > $O
for cfg in final-13682 dubrovnik-356; do
  for mode in fused nofuse host; do
    case $mode in
      fused)  export MIBA_ALLREDUCE=peer; unset MIBA_TAIL_NO_FUSE ;;
      nofuse) export MIBA_ALLREDUCE=peer; export MIBA_TAIL_NO_FUSE=1 ;;
      host)   export MIBA_ALLREDUCE=host; unset MIBA_TAIL_NO_FUSE ;;
    esac
    timeout -k 10 300 python3 -u $R/bench.py --gpus $N --single-process --same-device --config $cfg --steps 10 --warmup 2 --no-exact --cpu-iters 0 \
      --no-other-states --ttt 0 > /tmp/ab_line.json 2> /tmp/ab_err.txt || { tail -5 /tmp/ab_err.txt; exit 1; }
    python3 - "$cfg" "$mode" "$N" >> $O <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
print(json.dumps({"config": sys.argv[1], "mode": sys.argv[2], "ranks_on_one_device": int(sys.argv[3]), "lm_it_per_s": d["value"],
                  "ms_per_pcg_iteration": d["ms_per_pcg_iteration"], "pcg_iterations": d["pcg"]["pcg_iterations"],
                  "launches_per_iteration": d["pcg"]["launches_per_iteration"], "exchange": d["pcg"]["exchange"],
                  "product_ms": d["pcg"]["product_ms"], "tail_ms": d["pcg"]["tail_ms"]}))
PY
  done
done
cat $O
