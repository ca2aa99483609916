#!/bin/bash
# sweep window-kernel tuning knobs on cfg2 (diagnostic)
for R in 8 5; do for L in 0 32768 40960 53248 65536; do for N in 0 7 13; do
  echo -n "R=$R LDS=$L NCH=$N : "
  SDPGPU_WIN_R=$R SDPGPU_WIN_LDS=$L SDPGPU_WIN_NCH=$N python bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3e %.3f ms/step %.2f us/period' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']*1e3))"
done; done; done
