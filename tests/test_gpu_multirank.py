"""Two real processes, each driving its own slab through the HIP engine on the SAME GPU, exchange
their key / value rows every period (gloo, host-staged: RCCL cannot put two ranks on one device).
bench.py --check compares the sharded tables with a single-rank sweep bit for bit."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


W2 = ["--workload", "cfg2", "--weak"]
W5 = ["--workload", "cfg5", "--weak", "--states", "100000"]


# (the weak-scaling / K-period schedules of round 1 keep one case each: plain, interior / boundary split, calibration, blocked;
# SDP_MULTIRANK_ALL=1 runs their large-slab twins too -- every case is a two-process torch.distributed.run of bench.py,
# ~4 s of start-up each, and the GPU test run should stay within minutes)
_WEAK = [(W2 + ["--periods", "6"], "cfg2_small_slabs_key_rows"), (W2 + ["--periods", "6", "--split"], "cfg2_interior_boundary_split"),
         (W2 + ["--periods", "6", "CALIBRATE"], "cfg2_schedule_calibration"),
         (W5 + ["--periods", "4", "--schedule", "blocked2"], "f1_large_two_periods_per_exchange")]
if os.environ.get("SDP_MULTIRANK_ALL"):
    _WEAK += [(W5 + ["--periods", "3"], "f1_large_slabs"), (W5 + ["--periods", "3", "--split"], "f1_large_interior_boundary_split"),
              (W2 + ["--periods", "7", "--schedule", "blocked3"], "cfg2_three_periods_per_exchange")]
_STRONG = [(["--periods", "2"], "strong_target_grid"), (["--workload", "cfg2", "--periods", "5"], "strong_cfg2"),
           (["--workload", "cfg3", "--periods", "2"], "strong_cfg3"), (["--workload", "cfg3t", "--periods", "2"], "strong_cfg3_tenths"),
           (["--workload", "cfg4", "--periods", "3"], "strong_cfg4"), (["--workload", "cfg4p", "--periods", "2"], "strong_cfg4_pipeline"),
           (["--workload", "cfg5", "--states", "3000000", "--periods", "2"], "strong_cfg5_reduced_width")]


@pytest.mark.parametrize("extra", [e for e, _ in _WEAK + _STRONG], ids=[i for _, i in _WEAK + _STRONG])
def test_two_ranks_match_single_rank(extra):
    env = dict(os.environ)
    if "CALIBRATE" in extra:  # the N > 1 schedule calibration bench.py runs under RCCL, rehearsed over gloo
        extra = [e for e in extra if e != "CALIBRATE"]
        env["SDP_BENCH_CALIBRATE"] = "1"
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--backend", "gloo", "--check", "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["check_vs_single_rank"] is True
    assert rec["scaling"] == ("weak" if "--weak" in extra else "strong") and rec["value"] > 0
    assert rec["parity_gate"]["status"] == "ok" and rec["parity_gate"]["ranks"] == 2
    assert len(rec["config"]["cells_per_rank"]) == 2 and sum(rec["config"]["cells_per_rank"]) == rec["config"]["cells_per_step"]
    if "SDP_BENCH_CALIBRATE" in env:
        assert "calibrated" in rec["config"]["schedule"]


@pytest.mark.parametrize("extra", [["--periods", "3"], ["--workload", "cfg2", "--periods", "8"], ["--workload", "cfg3", "--periods", "2"],
                                   ["--workload", "cfg2", "--periods", "6", "--schedule", "overlap"]],
                         ids=["target", "cfg2_key_rows", "cfg3", "cfg2_overlapped"])
def test_native_sharded_path_with_a_world_of_one(extra):
    """bench.py's N > 1 branch on one GPU with a world of ONE: the gloo control group, the one-rank RCCL communicator
    created inside libsdpgpu.so (sdpgpu_comm_init), sdpgpu_solve_sharded (blocking / overlapped, calibrated), the per-rank
    parity gate, the per-rank reductions and the JSON line -- everything one rank of eight executes except a peer."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--rehearse-sharded",
           "--check", "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["check_vs_single_rank"] is True and rec["parity_gate"]["status"] == "ok"
    assert "RCCL all-gather issued by libsdpgpu.so" in rec["config"]["exchange"]
    assert rec["config"]["cells_per_rank"] == [rec["config"]["cells_per_step"]]


def test_bare_command_starts_its_own_ranks_host_exchange():
    """`python3 bench.py --gpus 2 --exchange host` with no launcher: bench.py starts its two ranks itself (self_launch: a child
    torch.distributed.run, never an exec) and relays exactly one JSON line with n_gpus = 2."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--exchange", "host",
           "--check", "--no-cpu-baseline", "--periods", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["check_vs_single_rank"] is True and rec["parity_gate"]["status"] == "ok"
    assert "host-staged" in rec["config"]["exchange"]


def _device_count():
    import torch
    return torch.cuda.device_count()  # (counts devices without initialising the GPU in this process)


@pytest.mark.parametrize("extra", [["--periods", "3"], ["--workload", "cfg2", "--periods", "8"]], ids=["target", "cfg2_key_rows"])
def test_two_processes_native_rccl_between_devices(extra):
    """The default N > 1 path itself: two processes, two DEVICES, ncclCommInitRank inside libsdpgpu.so and the in-place
    all-gather between them (--exchange native), checked against a single-rank sweep.  Needs a node with two GPUs: the
    one-GPU test box skips it (RCCL refuses two ranks on one device), a multi-GPU node runs it."""
    if _device_count() < 2:
        pytest.skip("needs two GPUs (RCCL between devices)")
    port = 29950 + os.getpid() % 40
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--exchange", "native", "--check", "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2000:])
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["check_vs_single_rank"] is True and rec["parity_gate"]["status"] == "ok"
    assert "RCCL all-gather issued by libsdpgpu.so" in rec["config"]["exchange"] and "fell back" not in rec["config"]["exchange"]
    assert len(rec["config"]["exchange_ms_per_rank"]) == 2 and rec["config"]["communicator_init_s"] > 0


def test_stalled_rank_ends_the_gpu_bench_with_a_record():
    """bench.py N > 1 on the GPU (two ranks over gloo on one device) with rank 1 stalled before its first sweep: both ranks'
    watchdogs print the one-line record and the launcher returns non-zero in well under two minutes."""
    port = 29900 + os.getpid() % 40
    env = dict(os.environ, SDP_WATCHDOG_INJECT_STALL="first sweep:1", SDP_WATCHDOG_SCALE="0.05")  # 300 s -> 15 s
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline", "--workload", "cfg2", "--periods", "4", "--weak"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0
    recs, dec = [], json.JSONDecoder()
    for l in out.stdout.splitlines():  # (a line may hold more than one record if two ranks' writes met in the launcher's pipe)
        pos = 0
        while l.startswith("{", pos) and "phase deadline exceeded" in l[pos:]:
            rec, pos = dec.raw_decode(l, pos)
            recs.append(rec)
    assert recs and all(r["phase"] == "first sweep" and r["value"] is None for r in recs)
    assert {r["rank"] for r in recs} <= {0, 1} and all(r["world"] == 2 for r in recs)
