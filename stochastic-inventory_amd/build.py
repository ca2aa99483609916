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
    return [os.path.join(CSRC, f) for f in ("sdpgpu.hip", "sdpgpu_generic.hip", "sdpgpu_window.hip", "sdpgpu_cash.hip",
                                            "sdpgpu_staff.hip", "sdpgpu_sparse.hip", "sdpgpu_comm.hip")]


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
    """Compile every translation unit (in parallel) and link libsdpgpu.so.  Only the entry points of
    include/sdpgpu.h are exported (-fvisibility=hidden + the visibility pragma in that header)."""
    if not force and up_to_date():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(HERE, "_build")
    os.makedirs(objdir, exist_ok=True)
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-fvisibility=hidden", *extra_flags]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc, *compile_flags, "-c", "-o", obj, src]
        if verbose:
            print("+", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs, "-lhiprtc"]
    if verbose:
        print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
