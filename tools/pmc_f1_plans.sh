#!/bin/bash
# VALU / LDS / wave counters of the F1 window kernel under two plans of configs[1]: the planner's (8,2,25 chunks) and a
# finer one (4,2,50 chunks: twice the waves, half the work each) that the cost model expects to win and does not.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for plan in "auto" "4 2 50"; do
  set -- $plan
  if [ "$1" = "auto" ]; then unset SDPGPU_WIN_R SDPGPU_WIN_S SDPGPU_WIN_NCH; tag=auto; else export SDPGPU_WIN_R=$1 SDPGPU_WIN_S=$2 SDPGPU_WIN_NCH=$3; tag=r$1s$2n$3; fi
  for grp in "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    g=$(echo $grp | tr ' ' '+')
    timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f1p/$tag/$g -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-target-grid > /dev/null 2>&1 || exit 1
  done
done
python3 - <<'PY'
import csv, glob, os, collections
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_f1p")
for tag in sorted(os.listdir(root)):
    agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for f in glob.glob(f"{root}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "window_f1_kernel" in r["Kernel_Name"] and "true, true>" in r["Kernel_Name"].split("(")[0]:
                a = agg[r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("==", tag)
    for k, (v, n, us) in sorted(agg.items()):
        print(f"  {k}: {v / max(n, 1):.5g} per launch ({n} launches, avg {us / max(n, 1):.1f} us under the profiler)")
PY
