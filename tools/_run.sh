for w in target cfg2 cfg4 cfg4p cfg3 cfg3t; do
  bash tools/pmc_collect.sh r02 $w > gpurun_out/collect_$w.log 2>&1 || echo "collect $w failed"
  echo "done $w"; head -3 gpurun_out/prof_r02_$w/r02_${w}_summary.txt
done
