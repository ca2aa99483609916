"""In-tree build of the HIP extension (gfx950 only) -- `python stochastic-inventory_amd/build.py`.

hipcc cross-compiles without a GPU.  The resulting libsdpgpu.so stays next to this file
(git-ignored, but it travels with the tree to the GPU box).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsdpgpu.so")

# -ffp-contract=off is part of the numerics contract (no FMA: Java fp64 semantics), not a tuning flag.
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-fPIC",
    "-shared",
    "-Wall",
    "-Wno-unused-function",
]


def sources():
    return [os.path.join(CSRC, "sdpgpu.hip"), os.path.join(CSRC, "sdpgpu_sparse.hip")]


def deps():
    out = [os.path.join(HERE, "..", "include", "sdpgpu.h")]
    for f in os.listdir(CSRC):
        out.append(os.path.join(CSRC, f))
    return out


def up_to_date() -> bool:
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(f) <= t for f in deps())


def build(force: bool = False, extra_flags=(), verbose: bool = True) -> str:
    if not force and up_to_date():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *HIPCC_FLAGS, *extra_flags, "-o", OUT, *sources(), "-lhiprtc"]
    if verbose:
        print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
