#!/usr/bin/env python3
"""Sweep forced plans of the F1 window kernel on the target grid (GPU box): one bench.py child per plan.

    python tools/f1_plan_sweep.py [--states N] [--workload target] R,S,NCH [R,S,NCH ...]   (0 = planner's choice)

Every child runs the parity gate (a smaller one than the default bench) before it is timed; a line per plan:
ms per sweep, cells/s, fraction of the fp64 issue bound, gate status, and the plan the library reports.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    states, workload, extra = 0, "target", []
    plans = []
    i = 0
    while i < len(args):
        if args[i] == "--states":
            states = int(args[i + 1]); i += 2
        elif args[i] == "--workload":
            workload = args[i + 1]; i += 2
        elif args[i] == "--periods":
            extra += ["--periods", args[i + 1]]; i += 2
        else:
            plans.append(tuple(int(v) for v in args[i].split(","))); i += 1
    for r, s, nch in plans:
        env = dict(os.environ, SDPGPU_DEBUG_PLAN="1")
        for k, v in (("SDPGPU_WIN_R", r), ("SDPGPU_WIN_S", s), ("SDPGPU_WIN_NCH", nch)):
            if v:
                env[k] = str(v)
            else:
                env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "5", "--warmup", "1",
               "--no-cpu-baseline", "--no-secondary", "--gate-cells", "1e9", *extra]
        if states:
            cmd += ["--states", str(states)]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        plan = [l for l in p.stderr.splitlines() if "window plan" in l]
        head = f"== states={states or 'default'} R={r} S={s} NCH={nch}"
        if p.returncode != 0:
            tail = (p.stderr.strip().splitlines() or ["?"])[-1]
            print(head, "rc", p.returncode, tail[:300], flush=True)
            continue
        rec = json.loads(p.stdout.strip().splitlines()[-1])
        print(head, f"{rec['ms_per_step']:.3f} ms  {rec['value']:.4e} cells/s  frac {rec['roofline']['frac']:.4f}  gate {rec['parity_gate']['status']}",
              "|", plan[-1].split("window plan:")[-1].strip() if plan else "", flush=True)


if __name__ == "__main__":
    main()
