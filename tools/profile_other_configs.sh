#!/bin/bash
# rocprofv3 kernel statistics for the bench workloads other than the default, and for the workforce driver.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/other
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in cfg3 cfg4 cfg4p; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$w.json 2> $OUT/$w.err
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/staff -- python3 $R/tools/workforce_drivers.py --no-check > $OUT/staff.log 2>&1
ls $OUT/*/*/ | head -30
