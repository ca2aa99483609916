timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "block_plans or states_per_lane" > gpurun_out/f2plans.log 2>&1; tail -12 gpurun_out/f2plans.log
