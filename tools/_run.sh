SDPGPU_CASH_DIAG_CHECK=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_xr.py tests/test_gpu_fuzz.py tests/test_gpu_big_grid.py tests/test_gpu_sharded_native.py -m gpu -x -q > gpurun_out/cash_tests.log 2>&1; tail -3 gpurun_out/cash_tests.log
for w in cfg3 cfg3t; do
  bash tools/pmc_collect.sh r02 $w > gpurun_out/collect_$w.log 2>&1 || echo "collect $w failed"
  head -2 gpurun_out/prof_r02_$w/r02_${w}_summary.txt
done
python tools/reference_drivers.py > gpurun_out/ref_drivers.txt 2>&1; grep -i "CashConstraint" gpurun_out/ref_drivers.txt | cut -c1-200
S=$(date +%s); python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench wall $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_default.json') if l.startswith('{')][-1])
def line(e,name):
    r=e['roofline']; print(name,'%.4g'%e['value'],'%.3f ms'%e['ms_per_step'],e['parity_gate']['status'],r.get('bound'),None if r.get('frac') is None else round(r['frac'],3),{k:round(v['frac'],3) for k,v in r.get('units',{}).items()})
line(d,d['config']['workload'])
for e in d.get('secondary',[]): line(e,e['workload'])
print(d['cpu_baseline'])
PY
