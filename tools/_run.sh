SDPGPU_CASH_DIAG_CHECK=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_big_grid.py tests/test_gpu_sharded_native.py -m gpu -x -q -k "diag or cfg3 or cash or dyadic" > gpurun_out/diag_tests.log 2>&1; tail -5 gpurun_out/diag_tests.log
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],d['config']['workload'],'%.4g'%d['value'],'%.3f'%d['ms_per_step'],d['parity_gate']['status'],r.get('bound'),r.get('frac'), r['per_launch_ms_events'])" $1 "$2"; }
timeout -k 10 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/c3_dma.json 2> gpurun_out/c3_dma.err && show gpurun_out/c3_dma.json dma
SDPGPU_CASH_DIAG_S=2 timeout -k 10 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/c3_dma2.json 2> gpurun_out/c3_dma2.err && show gpurun_out/c3_dma2.json dmaS2
