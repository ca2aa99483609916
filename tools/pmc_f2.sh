#!/bin/bash
# LDS counters of the F2 row-window kernel on configs[3]: are its ds_read_b64 with a 16-byte lane stride bank-conflicted?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '+')
  timeout -k 10 250 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f2/$tag -- python3 $R/bench.py --workload ${1:-cfg4} --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_f2_$tag.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, os, collections
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_f2")
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if ("window_f2_kernel" in r["Kernel_Name"] or "window_f1_kernel" in r["Kernel_Name"]) and "true>" in r["Kernel_Name"].split("(")[0]:
            a = agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in agg.items():
        print(f"{k[0]} {k[1]}: {v / max(n, 1):.5g} per launch over {n} launches")
PY
