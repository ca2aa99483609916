#!/bin/bash
# The counter passes of tools/pmc_collect.sh over ANY python program of this repo (kernels that are not bench workloads:
# the staff family's drivers, the user-functor kernel, the generic kernel):
#   bash tools/pmc_any.sh <round> <tag> tools/workforce_drivers.py --no-check
# Output: gpurun_out/prof_<round>_<tag>/ (kernel stats, one directory per counter pass, <round>_<tag>_summary.txt).
set -e
RND=$1; TAG=$2; shift 2
R=$GRAFT_REPO_ROOT
TOP=$R/gpurun_out/prof_${RND}_$TAG
OUT=$TOP/run_$(date +%Y%m%d%H%M%S)  # (one directory per collection: see pmc_collect.sh)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/$1 "${@:2}" > $OUT/stats.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $R/$1 "${@:2}" > $OUT/pmc$i.log 2>&1 || echo "pass $i ($grp) failed" >> $OUT/failed.txt
done
python3 $R/tools/pmc_reduce.py $OUT $RND $TAG
cp $OUT/${RND}_* $TOP/ 2>/dev/null
f=$(ls $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $TOP/kernel_stats.csv
true
