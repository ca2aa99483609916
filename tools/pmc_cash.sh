#!/bin/bash
# L1 / L2 counters of the F3 uniform-shift kernel on configs[2] (one pass per counter group; rocprofv3 directly on python3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  tag=$(echo $grp | tr ' ' '+')
  timeout -k 10 250 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cash/$tag -- python3 $R/bench.py --workload cfg3 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_cash_$tag.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, os, collections
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_cash")
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "cash_shift_kernel" in r["Kernel_Name"] and "Lb0E" in r["Kernel_Name"].split("cash_shift_kernelILb1E")[-1][:5]:
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in agg.items():
        print(f"{k}: {v / max(n, 1):.4g} per launch over {n} launches")
PY
