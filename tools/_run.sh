for v in 4; do
  SDPGPU_STAFF_WIN=$v timeout -k 10 600 python -m pytest tests/test_gpu_staff.py -m gpu -x -q > gpurun_out/staff_tests_$v.log 2>&1; tail -2 gpurun_out/staff_tests_$v.log
  SDPGPU_STAFF_WIN=$v timeout -k 10 600 python tools/workforce_drivers.py > gpurun_out/staff_drv_$v.log 2>&1; grep "WorkforceTesting" gpurun_out/staff_drv_$v.log | head -3
  SDPGPU_LIB=$PWD/_variants/sw3/libsdpgpu.so SDPGPU_STAFF_WIN=$v timeout -k 10 600 python tools/workforce_drivers.py > gpurun_out/staff_drv_w3_$v.log 2>&1; grep "WorkforceTesting" gpurun_out/staff_drv_w3_$v.log | head -3
done
