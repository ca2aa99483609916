python -m pytest tests -m gpu -x -q > gpurun_out/all_tests.log 2>&1; tail -3 gpurun_out/all_tests.log
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2],d['config']['workload'],'%.4g'%d['value'],'%.3f'%d['ms_per_step'],d['parity_gate']['status'],'%.3f'%d['roofline']['frac'])" $1 "$2"; }
for w in cfg4 cfg4p; do
  python bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/f2_$w.json 2> gpurun_out/f2_$w.err && show gpurun_out/f2_$w.json default
done
SDPGPU_CASH_PAIR_S=2 python bench.py --workload cfg3t --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/c3t_s2.json 2> gpurun_out/c3t_s2.err && show gpurun_out/c3t_s2.json pairS2
python bench.py --workload cfg3t --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/c3t_s1.json 2> gpurun_out/c3t_s1.err && show gpurun_out/c3t_s1.json pairS1
