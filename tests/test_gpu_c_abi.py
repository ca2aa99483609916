"""A C program (gcc, no C++/Python in the caller) drives the boundary the way CLSP.main drives the
reference, and must print the oracle's numbers."""
import os
import subprocess

import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_caller_reproduces_clsp_main(tmp_path, oracle, sia):
    w = cases.f1_clsp_main()
    pmf_file = tmp_path / "pmf.txt"
    with open(pmf_file, "w") as f:
        f.write(f"{w.T}\n")
        for tile in w.pmf:
            f.write(f"{len(tile)}\n")
            for d, p in tile:
                f.write(f"{float(d)!r} {float(p)!r}\n")
    exe = tmp_path / "c_abi_demo"
    libdir = os.path.join(ROOT, "stochastic-inventory_amd")
    subprocess.run(["gcc", "-O1", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", str(exe),
                    os.path.join(ROOT, "tests", "c_abi_demo.c"), "-L", libdir, "-lsdpgpu", f"-Wl,-rpath,{libdir}"],
                   check=True)
    out = subprocess.run([str(exe), str(pmf_file)], check=True, capture_output=True, text=True).stdout
    lines = dict(l.rsplit(" ", 1) for l in out.strip().splitlines() if " " in l)
    m = oracle.Problem(w.desc(), w.pmf).memo()
    assert float(lines["final optimal expected value is:"]) == m["value"]
    assert float(lines["optimal order quantity in the first priod is :"]) == m["action"]
    assert int(lines["cells"]) == 601 * 61 * sum(len(t) for t in w.pmf)


def _build_demo2(tmp_path):
    exe = tmp_path / "c_abi_demo2"
    libdir = os.path.join(ROOT, "stochastic-inventory_amd")
    subprocess.run(["gcc", "-O1", "-std=c11", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", str(exe),
                    os.path.join(ROOT, "tests", "c_abi_demo2.c"), "-L", libdir, "-lsdpgpu", f"-Wl,-rpath,{libdir}"],
                   check=True)
    return str(exe)


def test_c_caller_workforce(tmp_path, sia):
    """sdpgpu_set_level_pmf from plain C: WorkforcePlanning.main's costs on staff 0..30, hires 0..20, T = 3."""
    from oracle import staffref
    from stochastic_inventory_amd.pmf import staff_level_pmf
    T, rows = 3, 31
    table = staff_level_pmf([0.5] * T, rows)
    path = tmp_path / "staff.txt"
    with open(path, "w") as f:
        f.write(f"{T} {rows} 20 30\n")
        for t in range(T):
            for y in range(rows):
                f.write(" ".join(repr(float(v)) for v in table[t, y, : y + 1]) + "\n")
    out = subprocess.run([_build_demo2(tmp_path), "staff", str(path)], check=True, capture_output=True, text=True).stdout.splitlines()
    V, pol, _ = staffref.Problem(T=T, min_x=0, max_x=30, clamp=True, ini_x=0, max_hire=20, fix_cost=100, unit_vari_cost=10,
                                 salary=20, unit_penalty=80, min_staff=[8] * T, prob=table).solve()
    assert float(out[0].rsplit(" ", 1)[1]) == V[0][0] and int(out[1].rsplit(" ", 1)[1]) == pol[0][0]


def test_c_caller_two_product(tmp_path, sia, oracle):
    """sdpgpu_multicash_solve + sdpgpu_multi_set_table from plain C, against the oracle's memo."""
    import multicash_cases
    kw = multicash_cases.random_instance(7)
    path = tmp_path / "multicash.txt"
    with open(path, "w") as f:
        f.write(f"{kw['T']} {kw['q_bound']}\n")
        f.write(" ".join(repr(float(v)) for v in kw["price"] + kw["vari_cost"] + kw["sal_price"]) + "\n")
        f.write(" ".join(repr(float(kw[n])) for n in ("ini_cash", "ini_i1", "ini_i2", "min_inventory", "max_inventory",
                                                        "min_cash", "max_cash", "discount")) + "\n")
        for tile in kw["pmf"]:
            f.write(f"{len(tile)}\n")
            for d1, d2, p in tile:
                f.write(f"{float(d1)!r} {float(d2)!r} {float(p)!r}\n")
    out = subprocess.run([_build_demo2(tmp_path), "multicash", str(path)], check=True, capture_output=True, text=True).stdout.splitlines()
    (fv, q1, q2, states, cells), memo = oracle.memo_table("multicash", **kw)
    assert float(out[0].rsplit(" ", 1)[1]) == fv
    assert out[1].endswith(f"Q1 = {q1}, Q2 = {q2}")
    parts = out[2].replace(",", "").split()
    assert int(parts[2]) == sum(states) and int(parts[4]) == cells
    assert abs(float(parts[-1]) - memo[:, 6].sum()) <= 1e-9 * max(1.0, abs(memo[:, 6].sum()))  # (row order differs)


def _build_sharded(tmp_path):
    exe = tmp_path / "c_abi_sharded"
    libdir = os.path.join(ROOT, "stochastic-inventory_amd")
    subprocess.run(["gcc", "-O1", "-std=c11", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", str(exe),
                    os.path.join(ROOT, "tests", "c_abi_sharded.c"), "-L", libdir, "-lsdpgpu", f"-Wl,-rpath,{libdir}", "-lm"],
                   check=True)
    return str(exe)


@pytest.mark.parametrize("shape", [(1000, 40, 30, 5, 3), (20000, 64, 48, 4, 4), (130000, 24, 20, 3, 2)])
def test_c_caller_sharded_solves(tmp_path, sia, shape):
    """The collective behind the C ABI: RCCL communicator of one rank (sdpgpu_comm_init + sdpgpu_solve_sharded,
    blocking and overlapped), ncclCommInitAll (sdpgpu_solve_multi, one handle), and N rank-handles on this device --
    all bit-identical to sdpgpu_solve, from a C program (no Python, no torch in the process).  The small shape runs
    on key rows (several tasks per tile), the large ones on fp64 rows."""
    S, A, D, T, n = shape
    r = subprocess.run([_build_sharded(tmp_path), str(S), str(A), str(D), str(T), str(n)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert r.stdout.strip().splitlines()[-1].split() == ["ok", str(S * A * D * T)]  # (RCCL prints a version banner first)
