#!/bin/bash
# PMC passes over the product kernel alone (tools/time_k3.py), run on the GPU box: bash tools/pmc_k3.sh <tag> [ENV=value ...]
# Writes gpurun_out/pmc_<tag>/pmc.json (merged by tools/pmc_to_json.py).
set -o pipefail
tag=$1; shift
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PCG_ITS=12
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/p2 -o p -- python3 -u $R/tools/time_k3.py final-13682 5 > $O/p2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $O/p3 -o p -- python3 -u $R/tools/time_k3.py final-13682 5 > $O/p3.log 2>&1 || exit 1
if [ -z "$PMC_NO_TRAFFIC" ]; then
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p1 -o p -- python3 -u $R/tools/time_k3.py final-13682 5 > $O/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p0 -o p -- python3 -u $R/tools/time_k3.py final-13682 5 > $O/p0.log 2>&1 || exit 1
fi
cd $R
PMC_SOURCE="rocprofv3 --pmc passes over tools/time_k3.py final-13682 with 12 PCG iterations ($tag: $*)" \
  python3 tools/pmc_to_json.py $O/pmc.json $(find $O/p0 $O/p1 $O/p2 $O/p3 -name "*counter_collection.csv" 2>/dev/null) > $O/pmc_summary.txt 2>&1
python3 - <<PY
import json
d=json.load(open("$O/pmc.json"))
for k,v in d.items():
    if 'matvec' in k and isinstance(v,dict):
        print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
PY
