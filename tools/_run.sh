timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "states_per_lane" > gpurun_out/f2s.log 2>&1; tail -5 gpurun_out/f2s.log
