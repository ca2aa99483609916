timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "wide" > gpurun_out/fuzz_wide.log 2>&1; tail -5 gpurun_out/fuzz_wide.log
SDP_FUZZ_N=150 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "wide" > gpurun_out/fuzz_wide150.log 2>&1; tail -5 gpurun_out/fuzz_wide150.log
