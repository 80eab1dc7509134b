#!/bin/bash
# PMC passes over the exact solve (bench.py's exact_solve block at Final-13682): the assembly kernel k_dense_blocks and the tile kernels.
# Run on the GPU box: bash tools/pmc_exact.sh ; writes gpurun_out/pmc_exact/{summary.txt,pmc.json}
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_exact
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 -u $R/bench.py --steps 1 --warmup 0 --cpu-iters 0 --no-other-states --ttt 0"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p1 -o p -- $B > $O/p1.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p0 -o p -- $B > $O/p0.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/p2 -o p -- $B > $O/p2.log 2>&1 || exit 1
cd $R
PMC_SOURCE="rocprofv3 --pmc, three passes over bench.py --steps 1 --warmup 0 (exact_solve block: four LM iterations with SPARSE_SCHUR at Final-13682)" \
  python3 tools/pmc_to_json.py $O/pmc.json $(find $O/p0 $O/p1 $O/p2 -name "*counter_collection.csv") > $O/summary.txt 2>&1
grep "k_dense_blocks\|k_tchol" $O/summary.txt
