#!/bin/bash
# Round-3 profiles (run on the GPU box: bash tools/profile_r3.sh): kernel trace of the 5-step bench and four PMC passes over the
# product, the PCG tail and the Schur-init kernels at Final-13682.  Writes under gpurun_out/; the summaries are copied to profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -f $O/kernel_stats.csv ] || timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 -u $R/bench.py --steps 5 --warmup 2 --cpu-iters 0 --no-other-states --ttt 0 > $O/bench_steps5.json 2> $O/bench_steps5.err || exit 1
python3 $R/tools/kernel_stats_csv.py $O/trace/t_results.db $O/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --cpu-iters 0 --no-other-states --ttt 0   (round-3 build; exported from the rocpd database by tools/kernel_stats_csv.py: per-kernel totals over the whole process, i.e. set-up, 2 + 5 LM iterations with the PCG, and the five tile factorisations of bench.py's exact_solve block: k_tchol_*, k_dense_blocks)" || exit 1
python3 $R/tools/gap_analysis.py $O/trace/t_results.db > $O/gaps.txt 2>&1
export PCG_ITS=12
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p1 -o p -- python3 -u $R/tools/time_k3.py final-13682 5 > $O/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p0 -o p -- python3 -u $R/tools/time_k3.py final-13682 5 > $O/p0.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/p2 -o p -- python3 -u $R/tools/time_k3.py final-13682 5 > $O/p2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $O/p3 -o p -- python3 -u $R/tools/time_k3.py final-13682 5 > $O/p3.log 2>&1 || exit 1
cd $R
PMC_SOURCE="rocprofv3 --pmc, four passes (FETCH_SIZE | WRITE_SIZE | SQ busy, waves, VALU, LDS instructions | LDS index cycles, bank conflicts, instruction waits) over tools/time_k3.py final-13682 with 12 PCG iterations (round-3 build, default tiles)" \
  python3 tools/pmc_to_json.py $O/pmc.json $(find $O/p0 $O/p1 $O/p2 $O/p3 -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1
