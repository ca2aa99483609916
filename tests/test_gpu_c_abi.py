"""A C program (gcc, no C++/Python in the caller) drives the boundary the way CLSP.main drives the
reference, and must print the oracle's numbers."""
import os
import subprocess

import numpy as np
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
