"""The multi-PROCESS native path -- sdpgpu_comm_prepare / sdpgpu_comm_init with world > 1 in separate processes,
sdpgpu_solve_sharded's kernels and in-place all-gathers issued by libsdpgpu.so, bench.py's default `--exchange native` flow with
its watchdog phases -- run with 2 and 3 ranks on ONE GPU against a test double of RCCL (tests/mock_rccl/mock_rccl.c, loaded
through SDPGPU_RCCL_LIB: slabs travel through host shared memory).  RCCL itself refuses two ranks on one device, so until a
multi-GPU node runs tests/test_gpu_multirank.py::test_two_processes_native_rccl_between_devices this is as close as the
world > 1 branch of csrc/sdpgpu_comm.hip gets to being executed; what the double does not stand in for is RCCL's transport.
bench.py --check compares the sharded tables with a single-rank sweep, bit for bit."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.c")
LIB = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")


@pytest.fixture(scope="module")
def mock_rccl():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", LIB, SRC,
                        "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"], check=True)
    return LIB


CASES = [(2, ["--periods", "3"], "target_2_ranks"),
         (3, ["--workload", "cfg2", "--periods", "8"], "cfg2_key_rows_3_ranks"),
         (2, ["--workload", "cfg2", "--periods", "6", "--schedule", "overlap"], "cfg2_overlapped_second_stream"),
         (2, ["--workload", "cfg3t", "--periods", "2"], "cfg3t_ragged_action_counts"),
         (4, ["--workload", "cfg4", "--periods", "3"], "cfg4_4_ranks"),
         (3, ["--workload", "cfg5", "--states", "3000000", "--periods", "2"], "cfg5_reduced_width_3_ranks")]


@pytest.mark.parametrize("world,extra", [(w, e) for w, e, _ in CASES], ids=[i for _, _, i in CASES])
def test_native_exchange_between_processes(mock_rccl, world, extra):
    port = 29700 + (os.getpid() + world * 7 + len(extra)) % 200
    env = dict(os.environ, SDPGPU_RCCL_LIB=mock_rccl, MOCK_RCCL_SLOT_MB="64")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2",
           "--warmup", "1", "--exchange", "native", "--check", "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == world and rec["check_vs_single_rank"] is True
    assert rec["parity_gate"]["status"] == "ok" and rec["parity_gate"]["ranks"] == world
    assert "RCCL all-gather issued by libsdpgpu.so" in rec["config"]["exchange"] and "fell back" not in rec["config"]["exchange"]
    assert len(rec["config"]["cells_per_rank"]) == world and sum(rec["config"]["cells_per_rank"]) == rec["config"]["cells_per_step"]
    assert len(rec["config"]["exchange_ms_per_rank"]) == world and rec["config"]["communicator_init_s"] > 0
    assert "communicator init" in rec["config"]["phases_s"] and "timed loop" in rec["config"]["phases_s"]


def _bare_env(**extra):
    env = dict(os.environ, **extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("world,extra", [(2, ["--periods", "3"]), (4, ["--workload", "cfg2", "--periods", "6"])], ids=["target_2", "cfg2_4"])
def test_bare_command_starts_its_own_ranks_native(mock_rccl, world, extra):
    """`python3 bench.py --gpus N` with NO launcher (bench.py: self_launch): the process starts N ranks as a child, the default
    --exchange native path runs between them (mock transport), and exactly ONE JSON line comes back with n_gpus = N."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--check",
           "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=ROOT,
                         env=_bare_env(SDPGPU_RCCL_LIB=mock_rccl, MOCK_RCCL_SLOT_MB="64"))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == world and rec["check_vs_single_rank"] is True and rec["parity_gate"]["status"] == "ok"
    assert "RCCL all-gather issued by libsdpgpu.so" in rec["config"]["exchange"] and "fell back" not in rec["config"]["exchange"]
    assert "[bench phase] rank 0: timed loop" in out.stderr  # the ranks announced their phases to the parent


def test_bare_command_relays_a_stalled_rank_as_one_record(mock_rccl):
    """The bare command with rank 1 stalled before its first sweep: the ranks' watchdogs fire, the launcher ends, and the
    parent prints ONE record naming the phase and exits non-zero."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--workload", "cfg2", "--periods", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env=_bare_env(SDPGPU_RCCL_LIB=mock_rccl, SDP_WATCHDOG_INJECT_STALL="first sweep:1", SDP_WATCHDOG_SCALE="0.05"))
    assert out.returncode != 0
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["value"] is None and rec["phase"] == "first sweep" and rec["n_gpus"] == 2
    assert rec["rank_records"] and all(r["error"] == "phase deadline exceeded" for r in rec["rank_records"])


def test_a_rank_that_cannot_prepare_does_not_strand_its_peers(mock_rccl):
    """Rank 1's collective library does not load (SDPGPU_RCCL_LIB points nowhere on that rank): sdpgpu_comm_prepare fails
    THERE, the ranks agree before anybody enters ncclCommInitRank, and the run ends -- here with a non-zero exit, since the
    torch fallback cannot put two ranks on one GPU either -- instead of hanging rank 0 inside the collective."""
    port = 29980 + os.getpid() % 15
    env = dict(os.environ, SDPGPU_RCCL_LIB=mock_rccl, SDP_TEST_BREAK_RCCL_ON_RANK="1", SDP_WATCHDOG_SCALE="0.5")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "0", "--exchange", "native", "--no-cpu-baseline", "--workload", "cfg2", "--periods", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert "native RCCL communicator unavailable" in out.stderr and "cannot load SDPGPU_RCCL_LIB" in out.stderr
    assert "phase deadline exceeded" not in out.stdout  # nobody waited in a collective: the ranks agreed and moved on


_MULTI_DRIVER = r"""
import os, sys
sys.path.insert(0, os.environ["SDP_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SDP_ROOT"], "tests"))
import numpy as np
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
import cases
world, threads, name = int(sys.argv[1]), sys.argv[2] == "1", sys.argv[3]
w = {"cfg2": lambda: workloads.cfg2_clsp(T=6), "target": lambda: workloads.target_grid(T=3, S=300000),
     "cfg3t": lambda: workloads.cfg3_tenths(T=3, NX=40, maxCash=400.0, A=60, D=25), "f5": cases.f5_cash_leadtime}[name]()
d1 = w.desc(); d1.device = 0
ref = sia.SdpEngine(d1, w.pmf, w.overhead()); ref.solve(sync=True)
engs = []
for r in range(world):
    d = w.desc(); d.rank, d.world_size, d.device = r, world, 0
    engs.append(sia.SdpEngine(d, w.pmf, w.overhead()))
for rep in range(2):
    sia.SdpEngine.solve_multi(engs, sync=True, gather_first=True, threads=threads)
    for r, e in enumerate(engs):
        for period in range(1, w.T + 1):
            _, lo, hi = e.slab(period)
            assert np.array_equal(e.values(period), ref.values(period)), (name, r, period)
            assert np.array_equal(e.policy(period), ref.policy(period)[lo:hi]), (name, r, period)
print("MULTI_OK", world, threads, name)
"""


@pytest.mark.parametrize("threads", ["0", "1"], ids=["one-thread-grouped", "thread-per-rank"])
@pytest.mark.parametrize("world,name", [(4, "cfg2"), (8, "target"), (3, "cfg3t"), (2, "f5")])
def test_solve_multi_communicator_branch(mock_rccl, world, name, threads):
    """sdpgpu_solve_multi's COMMUNICATOR branch -- ncclCommInitAll, then per period every rank's kernel and one ncclGroupStart /
    ncclAllGather x n / ncclGroupEnd (default), or one host thread per rank each issuing its own all-gather
    (SDPGPU_SHARDED_THREADS) -- which on real hardware needs one device per rank: here all ranks on one GPU against the test
    double (SDPGPU_MULTI_EXCHANGE=rccl forces the branch), every table against the single-rank sweep, twice (the second
    call reuses the communicators)."""
    env = dict(os.environ, SDPGPU_RCCL_LIB=mock_rccl, SDPGPU_MULTI_EXCHANGE="rccl", MOCK_RCCL_SLOT_MB="16", SDP_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-c", _MULTI_DRIVER, str(world), threads, name], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0 and "MULTI_OK" in out.stdout, (out.stdout[-800:], out.stderr[-2500:])


_FAIL_DRIVER = r"""
import os, sys
sys.path.insert(0, os.environ["SDP_ROOT"])
import numpy as np
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
world, inject = int(sys.argv[1]), sys.argv[2]
w = workloads.cfg2_clsp(T=6)
d1 = w.desc(); d1.device = 0
ref = sia.SdpEngine(d1, w.pmf, w.overhead()); ref.solve(sync=True)
engs = []
for r in range(world):
    d = w.desc(); d.rank, d.world_size, d.device = r, world, 0
    engs.append(sia.SdpEngine(d, w.pmf, w.overhead()))
os.environ["SDPGPU_TEST_FAIL_RANK"] = inject
try:
    sia.SdpEngine.solve_multi(engs, sync=True, gather_first=True, threads=True)
    print("NO_ERROR")
except Exception as exc:
    print("FAILED_AS_EXPECTED:", exc)
del os.environ["SDPGPU_TEST_FAIL_RANK"]
# the call after the failure: new communicators, the right tables
sia.SdpEngine.solve_multi(engs, sync=True, gather_first=True, threads=True)
for r, e in enumerate(engs):
    for period in range(1, w.T + 1):
        _, lo, hi = e.slab(period)
        assert np.array_equal(e.values(period), ref.values(period)), (r, period)
        assert np.array_equal(e.policy(period), ref.policy(period)[lo:hi]), (r, period)
print("RECOVERED_OK")
"""


@pytest.mark.parametrize("inject", ["kernel:1:4", "collective:2:3", "kernel:0:6", "collective:0:1"])
def test_thread_per_rank_failure_does_not_strand_the_peers(mock_rccl, inject):
    """in sdpgpu_solve_multi's thread-per-rank communicator branch a rank that fails must not leave its peers inside
    an all-gather it never joins.  One rank is made to fail where its kernel of a period would be launched (the peers must not
    enqueue that period's collective: rendezvous before the all-gather) or where its all-gather would be enqueued (the peers ARE
    inside the collective -- the double blocks there as RCCL's kernel would -- and are released by ncclCommAbort): the call
    returns the failing rank's error, nobody hangs, and the next call solves the problem on fresh communicators."""
    env = dict(os.environ, SDPGPU_RCCL_LIB=mock_rccl, SDPGPU_MULTI_EXCHANGE="rccl", MOCK_RCCL_SLOT_MB="16", SDP_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-c", _FAIL_DRIVER, "3", inject], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-800:], out.stderr[-2500:])
    rank = inject.split(":")[1]
    assert "FAILED_AS_EXPECTED" in out.stdout and f"rank {rank}: injected failure" in out.stdout, out.stdout
    assert "RECOVERED_OK" in out.stdout
