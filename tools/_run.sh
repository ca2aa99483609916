SDPGPU_CASH_DIAG_CHECK=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "diag or cfg3 or cash" > gpurun_out/diag_tests.log 2>&1; tail -3 gpurun_out/diag_tests.log
for w in cfg3 cfg3t; do
  bash tools/pmc_collect.sh r02 $w > gpurun_out/collect_$w.log 2>&1 || echo "collect $w failed"
  head -3 gpurun_out/prof_r02_$w/r02_${w}_summary.txt
done
