"""Per-period launch times of the staff workload under the launcher's kernel choices (environment switches of sdpgpu_staff.hip)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for label, env in (("default", {}), ("pair + window S=4 everywhere", {"SDPGPU_STAFF_PAIR": "1", "SDPGPU_STAFF_WIN": "4"}),
                   ("pair + window S=2 everywhere", {"SDPGPU_STAFF_PAIR": "1", "SDPGPU_STAFF_WIN": "2"}),
                   ("pair kernel, no window", {"SDPGPU_STAFF_PAIR": "1", "SDPGPU_STAFF_WIN": "0"})):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "staff", "--no-secondary", "--no-cpu-baseline",
                          "--steps", "5", "--warmup", "2"], capture_output=True, text=True, env=dict(os.environ, **env))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        print(label, "FAILED", out.stderr[-300:]); continue
    r = json.loads(lines[-1])
    print(f"{label:32s} {r['value']:.3e} cells/s  {r['ms_per_step']:.3f} ms  gate {r['parity_gate']['status']}  per period (T..1): "
          + " ".join(f"{x:.2f}" for x in r["roofline"]["per_launch_ms_events"]), flush=True)
