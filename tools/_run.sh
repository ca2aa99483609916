SDPGPU_CASH_DIAG_CHECK=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/all_tests.log 2>&1; tail -4 gpurun_out/all_tests.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
def line(e,name):
    r=e['roofline']; print(name,'%.4g'%e['value'],'%.3f ms'%e['ms_per_step'],e['parity_gate']['status'],r.get('bound'),None if r.get('frac') is None else round(r['frac'],3),{k:round(v['frac'],3) for k,v in r.get('units',{}).items()})
line(d,d['config']['workload'])
for e in d.get('secondary',[]): line(e,e['workload'])
print(d['cpu_baseline']['value'])
PY
bash tools/pmc_collect.sh x5 cfg4 > /dev/null 2>&1; head -28 gpurun_out/prof_x5_cfg4/*summary.txt
