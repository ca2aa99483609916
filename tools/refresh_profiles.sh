#!/bin/bash
# The profiles of record for the bench default (configs[1]) from ONE build: bench line, rocprofv3 kernel stats, PMC traffic.
# Run on the GPU box from the repo root: bash tools/refresh_profiles.sh   (outputs under gpurun_out/refresh/)
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/refresh
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-target-grid > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-target-grid > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
done
ls $OUT/stats/*/ | head
