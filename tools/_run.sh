SDPGPU_CASH_DIAG_CHECK=1 timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/all_tests.log 2>&1; tail -6 gpurun_out/all_tests.log
