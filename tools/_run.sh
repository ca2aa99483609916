timeout -k 10 900 python -m pytest tests/test_gpu_big_grid.py -m gpu -x -q -k "cfg4_full" > gpurun_out/cfg4_full.log 2>&1; tail -3 gpurun_out/cfg4_full.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_default.json') if l.startswith('{')][-1])
def line(e,name):
    r=e['roofline']; print(name,'%.4g'%e['value'],'%.3f ms'%e['ms_per_step'],e['parity_gate']['status'],r.get('bound'),None if r.get('frac') is None else round(r['frac'],3))
line(d,d['config']['workload'])
for e in d.get('secondary',[]): line(e,e['workload'])
PY
