python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
/usr/bin/time -v python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; grep -E "Elapsed|Maximum resident" gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_default.json') if l.startswith('{')][-1])
def line(e,name):
    r=e['roofline']; print(name,'%.4g'%e['value'],'%.3f ms'%e['ms_per_step'],e['parity_gate']['status'],r.get('bound'),None if r.get('frac') is None else round(r['frac'],3),{k:round(v['frac'],3) for k,v in r.get('units',{}).items()}, r.get('counters'))
line(d,d['config']['workload'])
for e in d.get('secondary',[]): line(e,e['workload'])
print(d['cpu_baseline'])
PY
SDPGPU_CASH_DIAG_CHECK=1 timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/all_tests.log 2>&1; tail -4 gpurun_out/all_tests.log
