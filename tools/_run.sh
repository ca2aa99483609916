SDPGPU_CASH_DIAG_CHECK=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_big_grid.py -m gpu -x -q -k "diag or cfg3 or cash" > gpurun_out/diag_tests.log 2>&1; tail -5 gpurun_out/diag_tests.log
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2],d['config']['workload'],'%.4g'%d['value'],'%.3f'%d['ms_per_step'],d['parity_gate']['status'],d['roofline'].get('frac'), d['roofline']['per_launch_ms_events'])" $1 "$2"; }
timeout -k 10 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/c3_diag1.json 2> gpurun_out/c3_diag1.err && show gpurun_out/c3_diag1.json diagS1
