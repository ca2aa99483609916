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
                                            "sdpgpu_staff.hip", "sdpgpu_sparse.hip", "sdpgpu_comm.hip", "sdpgpu_pmf.hip")]


def deps():
    out = [os.path.join(HERE, "..", "include", "sdpgpu.h")]
    for f in os.listdir(CSRC):
        out.append(os.path.join(CSRC, f))
    return out


def source_sha() -> str:
    """Digest of everything the library is built from (tools/kernel_sha.py: build_source_sha)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_kernel_sha", os.path.join(HERE, "..", "tools", "kernel_sha.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_source_sha(os.path.join(HERE, ".."))


def baked_build_id(path: str = OUT):
    """The identity a built library carries (what sdpgpu_build_id returns), read from the file without loading it."""
    marker = b"sdpgpu-build-id:"
    try:
        blob = open(path, "rb").read()
    except OSError:
        return None
    at = blob.find(marker)
    if at < 0:
        return None
    end = blob.find(b"\0", at)
    return blob[at + len(marker):end].decode("ascii", "replace")


def up_to_date() -> bool:
    """The library exists and was built from exactly these sources (by content, not by modification time: the .so is a
    git-ignored artefact that travels with the tree)."""
    return os.path.exists(OUT) and baked_build_id() == source_sha()


def build(force: bool = False, extra_flags=(), verbose: bool = True) -> str:
    """Compile every translation unit (in parallel) and link libsdpgpu.so.  Only the entry points of
    include/sdpgpu.h are exported (-fvisibility=hidden + the visibility pragma in that header)."""
    if not force and up_to_date():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(HERE, "_build")
    os.makedirs(objdir, exist_ok=True)
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-fvisibility=hidden", f'-DSDPGPU_BUILD_ID="{source_sha()}"',
                                                                   *extra_flags]

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


ASAN_DIR = os.path.join(HERE, "_build_asan")
ASAN_OUT = os.path.join(ASAN_DIR, "libsdpgpu_hostasan.so")


def asan_runtime() -> str:
    """The shared AddressSanitizer runtime of ROCm's clang (LD_PRELOAD it to load the library below into Python)."""
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not hits:
        raise FileNotFoundError("libclang_rt.asan-x86_64.so not found under /opt/rocm/lib/llvm")
    return hits[-1]


def build_host_asan(force: bool = False, verbose: bool = False) -> str:
    """HOST-side AddressSanitizer + UBSan build of the same sources (-fno-gpu-sanitize: device code is compiled as
    usual; GPU ASan is not available on the pool).  It exists for the CPU checks of the host half of the library --
    descriptor validation, per-period layout, slab and halo arithmetic, the window planner, footprints, state
    indexing -- which tests/test_sanitizers.py drives through the C ABI without a GPU.  Not shipped, not loaded by
    the product."""
    if not force and os.path.exists(ASAN_OUT) and baked_build_id(ASAN_OUT) == source_sha():
        return ASAN_OUT
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(ASAN_DIR, exist_ok=True)
    san = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined", "-shared-libasan"]
    flags = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
             "-fvisibility=hidden", "-Wno-unused-function", f'-DSDPGPU_BUILD_ID="{source_sha()}"', *san]

    def compile_one(src):
        obj = os.path.join(ASAN_DIR, os.path.basename(src) + ".o")
        cmd = [hipcc, *flags, "-c", "-o", obj, src]
        if verbose:
            print("+", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(7, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *san, "-o", ASAN_OUT, *objs, "-lhiprtc"], check=True)
    return ASAN_OUT


if __name__ == "__main__":
    if "--host-asan" in sys.argv:
        print(build_host_asan(force="--force" in sys.argv, verbose=True))
    else:
        build(force="--force" in sys.argv)
