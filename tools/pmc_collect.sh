#!/bin/bash
# Profiles of record for ONE bench workload from ONE build (run on the GPU box from the repo root):
#   bash tools/pmc_collect.sh <round> <workload> [extra bench.py args]
# 1. the bench line itself (gated, un-profiled HIP-event times per launch),
# 2. rocprofv3 --kernel-trace --stats,
# 3. counter passes, each in its own run with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots):
#    FETCH_SIZE | WRITE_SIZE | SQ instruction mix | SQ busy/active cycles | TA / TCP / TCC
# and reduces them with tools/pmc_reduce.py into gpurun_out/prof_<round>_<workload>/ (copy the summaries to profiles/).
set -e
RND=$1; W=$2; shift 2
R=$GRAFT_REPO_ROOT
TOP=$R/gpurun_out/prof_${RND}_$W
# every collection in its own directory: gpurun MERGES gpurun_out/ back into the local copy, and a reduce over the union of
# two collections would average counters of different builds
OUT=$TOP/run_$(date +%Y%m%d%H%M%S)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --workload $W --no-secondary $*"
timeout -k 10 500 $BENCH --cpu-seconds 5 > $OUT/bench.json 2> $OUT/bench.err
PROF="--workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-gate $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $PROF > $OUT/stats.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py $PROF > $OUT/pmc$i.log 2>&1 || echo "pass $i ($grp) failed" >> $OUT/failed.txt
done
python3 $R/tools/pmc_reduce.py $OUT $RND $W
cp $OUT/${RND}_* $OUT/bench.json $TOP/ 2>/dev/null
f=$(ls $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $TOP/kernel_stats.csv
true
