#!/bin/bash
# cash_diag_kernel on configs[2], rows-per-XCD (SDPGPU_CASH_DIAG_BANDS=0, round 2) against cash bands per XCD (default, round 3):
# time per sweep, then fabric traffic and L2 hits from the counters.  -> gpurun_out/r03_diag_bands.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_diag_bands
{
for b in 0 1; do
  export SDPGPU_CASH_DIAG_BANDS=$b
  python3 $R/bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT.bench$b.json 2>/dev/null
  python3 -c "
import json;d=json.loads(open('$OUT.bench$b.json').read().strip().splitlines()[-1]);print('bands=$b: %.3f ms per sweep, %.4e cells/s, per-launch ms %s, gate %s' % (d['ms_per_step'], d['value'], d['roofline']['per_launch_ms_events'], d['parity_gate']['status']))"
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=b${b}_$(echo $grp | tr ' ' '+')
    timeout -k 10 250 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -- python3 $R/bench.py --workload cfg3 --periods 3 --steps 1 --warmup 0 --no-cpu-baseline --no-gate > $OUT.$tag.log 2>&1 || exit 1
  done
done
python3 - <<'PY'
import csv, glob, os, collections
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_diag_bands")
for b in (0, 1):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(root + f"/b{b}_*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if "cash_diag_kernel" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f"-- bands={b}: cash_diag_kernel, counters per launch")
    for k, (v, n) in sorted(agg.items()):
        print(f"   {k}: {v / max(n, 1):.5g} over {n} launches")
    g = lambda k: agg[k][0] / max(agg[k][1], 1)
    if g("FETCH_SIZE"):
        print(f"   fabric read bytes per launch (FETCH_SIZE KiB x 1024 x 2, gfx950 correction): {g('FETCH_SIZE') * 2048:.4g}")
    if g("TCC_HIT_sum"):
        print(f"   L2 hit rate: {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.4f}")
PY
} > $R/gpurun_out/r03_diag_bands.txt 2>&1
cat $R/gpurun_out/r03_diag_bands.txt
