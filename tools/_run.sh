timeout -k 10 900 python -m pytest tests/test_gpu_staff.py -m gpu -x -q > gpurun_out/staff_tests.log 2>&1; tail -15 gpurun_out/staff_tests.log
