#!/bin/bash
# Vector-L1 / L2 counters of cash_row_pair_kernel on CashConstraint.main's grid, one tile per workgroup (default) against
# the shared-block form (SDPGPU_CASH_SHARE=1): is the shared form's extra time L1 misses?  -> gpurun_out/r03_cash_share_l1.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_cash_share
for share in 0 1; do
  export SDPGPU_CASH_SHARE=$share
  for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES"; do
    tag=share${share}_$(echo $grp | tr ' ' '+')
    timeout -k 10 250 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -- python3 $R/bench.py --workload cfg3t --periods 3 --steps 1 --warmup 0 --no-cpu-baseline --no-gate > $OUT.$tag.log 2>&1 || exit 1
  done
done
python3 - <<'PY' > $R/gpurun_out/r03_cash_share_l1.txt
import csv, glob, os, collections
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_cash_share")
print("cash_row_pair_kernel<LAST=false> on cfg3t (periods before T), counters per launch; share0 = one tile per workgroup (default), share1 = SDPGPU_CASH_SHARE=1")
for share in (0, 1):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(root + f"/share{share}_*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if "cash_row_pair_kernel<false" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f"-- share{share}")
    for k, (v, n) in sorted(agg.items()):
        print(f"   {k}: {v / max(n, 1):.5g} per launch over {n} launches")
    g = lambda k: agg[k][0] / max(agg[k][1], 1)
    if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
        print(f"   vector-L1 miss share: {g('TCP_TCC_READ_REQ_sum') / g('TCP_TOTAL_CACHE_ACCESSES_sum'):.3f} of accesses go on to L2")
PY
cat $R/gpurun_out/r03_cash_share_l1.txt
