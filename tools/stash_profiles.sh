#!/bin/bash
# Copy the judged summaries of tools/pmc_collect.sh runs from gpurun_out/ (scratch) into profiles/ (tracked):
#   bash tools/stash_profiles.sh r02 target cfg2 ...
RND=$1; shift
for W in "$@"; do
  D=gpurun_out/prof_${RND}_$W
  [ -d $D ] || { echo "no $D"; continue; }
  cp $D/${RND}_pmc_*.json profiles/ 2>/dev/null
  cp $D/${RND}_${W}_summary.txt profiles/ 2>/dev/null
  cp $D/bench.json profiles/${RND}_${W}_bench.json 2>/dev/null
  [ -f $D/kernel_stats.csv ] && cp $D/kernel_stats.csv profiles/${RND}_${W}_kernel_stats.csv
done
ls profiles | grep "^$RND" | head -40
