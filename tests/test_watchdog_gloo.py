"""A stalled rank must end a multi-rank run with a diagnosis, not hang it (bench.py N > 1; stochastic_inventory_amd/watchdog.py).

Two `gloo` ranks walk through the phase sequence of bench.py's sharded path (process group -> communicator init -> first
sweep -> ...), each phase ending in a collective.  One rank is made to stall inside a phase (SDP_WATCHDOG_INJECT_STALL,
the hook bench.py itself honours): its own watchdog AND the watchdog of the rank waiting for it in the collective print a
one-line JSON record naming the phase and the rank, and both processes exit non-zero well inside a minute."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, time
sys.path.insert(0, os.environ["SDP_ROOT"])
import torch
import torch.distributed as dist
from stochastic_inventory_amd.watchdog import PhaseWatchdog

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
wd = PhaseWatchdog(rank, world, context={"metric": "(state,action,demand) cell evals/sec", "n_gpus": world})
with wd.phase("process group", 60):
    dist.init_process_group("gloo", rank=rank, world_size=world)
for name in ("communicator init", "first sweep", "parity gate", "schedule calibration", "timed loop"):
    with wd.phase(name, 4):
        time.sleep(0.05)                 # the phase's own work
        t = torch.ones(1)
        dist.all_reduce(t)               # ... and the collective that ends it
with wd.phase("teardown", 30):
    dist.barrier()
    dist.destroy_process_group()
wd.close()
print('{"value": 1.0, "phases": %d}' % len(wd.history), flush=True)
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, inject=""):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GLOO_SOCKET_IFNAME="lo", SDP_ROOT=ROOT)
        env.pop("SDP_WATCHDOG_INJECT_STALL", None)
        if inject:
            env["SDP_WATCHDOG_INJECT_STALL"] = inject
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    return procs


def _collect(procs, timeout):
    t0 = time.monotonic()
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=max(1.0, timeout - (time.monotonic() - t0)))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail(f"the run hung: a rank was still alive after {timeout} s")
        outs.append((p.returncode, out, err))
    return outs, time.monotonic() - t0


def test_healthy_run_passes_every_phase():
    outs, took = _collect(_launch(2), 120)
    for rc, out, err in outs:
        assert rc == 0, err[-2000:]
        assert json.loads(out.strip().splitlines()[-1])["phases"] == 7


@pytest.mark.parametrize("phase,stalled", [("first sweep", 1), ("communicator init", 0)])
def test_stalled_rank_ends_the_run_with_a_record(phase, stalled):
    outs, took = _collect(_launch(2, inject=f"{phase}:{stalled}"), 60)
    assert took < 60
    records = []
    for rank, (rc, out, err) in enumerate(outs):
        assert rc == 3, f"rank {rank}: rc {rc}\n{err[-2000:]}"
        rec = json.loads(out.strip().splitlines()[-1])           # the record is on stdout (where the bench line would be) ...
        assert json.loads([l for l in err.splitlines() if l.startswith("{")][-1]) == rec  # ... and on stderr
        assert rec["error"] == "phase deadline exceeded" and rec["phase"] == phase and rec["rank"] == rank
        assert rec["world"] == 2 and rec["value"] is None and rec["elapsed_s"] >= rec["deadline_s"] == 4
        assert [n for n, _ in rec["phases_done"]][0] == "process group"
        records.append(rec)
    assert len(records) == 2
