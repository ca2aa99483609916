"""`python bench.py --gpus N` with N > 1 and no launcher (bench.py: self_launch).  The process starts its N ranks as a child
`torch.distributed.run` (never an exec) and relays ONE JSON line; a failure of the ranks -- here, on a machine without a GPU,
every rank refuses before the process group exists -- ends the parent with a non-zero code and ONE JSON error record, inside
a minute; ranks that never come back are ended at --launch-timeout, again with a record.  (The healthy N = 2 runs of the bare
command are GPU tests: tests/test_gpu_multirank.py, tests/test_gpu_native_mock_rccl.py.)"""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpu_present():
    import torch
    return torch.cuda.device_count() > 0


def _bare(*extra, env=None, timeout=120):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", *extra], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)
    return out, time.monotonic() - t0


def test_ranks_failing_before_the_process_group_end_the_parent_with_a_record():
    if _gpu_present():
        pytest.skip("needs a machine WITHOUT a GPU: there the ranks refuse before the process group")
    out, took = _bare()
    assert out.returncode != 0 and took < 60
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout: the record"
    rec = json.loads(lines[0])
    assert rec["value"] is None and rec["error"] == "ranks exited with a non-zero code" and rec["phase"] == "launch"
    assert rec["n_gpus"] == 2 and rec["returncode"] != 0
    assert any("needs a GPU" in f for f in rec["fatal"]), rec


def test_ranks_that_never_come_back_are_ended_at_the_launch_timeout():
    out, took = _bare("--launch-timeout", "6", env={"SDP_BENCH_TEST_STALL_AT_START": "600"})
    assert out.returncode != 0 and took < 60
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["error"] == "launch timeout exceeded" and rec["value"] is None and rec["launch_timeout_s"] == 6.0
    # nothing of the process group the parent started is left behind
    time.sleep(0.5)
    left = subprocess.run(["ps", "-eo", "pid,args"], capture_output=True, text=True).stdout
    assert not [l for l in left.splitlines() if "bench.py" in l and "--launch-timeout 6" in l], left


def test_under_a_launcher_the_process_does_not_launch_again():
    """WORLD_SIZE in the environment = already a rank: no child is started (a world that contradicts --gpus is refused)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                         cwd=ROOT, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in out.stderr
    assert "torch.distributed.run" not in out.stderr
