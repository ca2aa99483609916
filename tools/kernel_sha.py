"""Digest of the kernel sources a bench workload runs on: ties a counter summary in profiles/ to the build it was measured
on (bench.py uses a summary only if the digest matches; tools/pmc_reduce.py writes it).  Per workload, so that work on
one kernel family does not retire the summaries of the others."""
import hashlib
import os

COMMON = ["sdp_device.hpp", "sdpgpu_internal.hpp"]
WINDOW = ["sdp_window.hpp", "sdpgpu_window.hip"]   # F1 / F2 window kernels: target, cfg2, cfg4, cfg4p, cfg5
CASH = ["sdp_cash.hpp", "sdpgpu_cash.hip"]         # uniform-shift, diagonal and cash row kernels: cfg3, cfg3t, f5_spl
STAFF = ["sdp_staff.hpp", "sdpgpu_staff.hip"]      # workforce.StaffRecursion's kernels
SPARSE = ["sdpgpu_sparse.hip"]                     # the reachable-set engine of the two-product families
CUSTOM = ["sdp_custom_src.hpp", "sdp_gather.hpp", "sdpgpu_generic.hip"]  # user lambdas (hipRTC) / generic period kernel


def kernel_files(workload_name: str):
    n = workload_name
    if n.startswith("cfg3") or n.startswith("f5_") or n.startswith("separable_f5"):
        fam = CASH
    elif n.startswith("staff"):
        fam = STAFF
    elif n.startswith("multilead") or n.startswith("multicash") or n.startswith("multixr"):
        return sorted(SPARSE)  # (a translation unit of its own: nothing of the grid kernels' headers)
    elif n.startswith("custom_clsp_level"):
        fam = CUSTOM + WINDOW  # (the user's cost functions tabulated, the F1 window kernel reading the tables)
    elif n.startswith("custom"):
        fam = CUSTOM
    else:
        fam = WINDOW  # target, cfg2, cfg4, cfg4p, cfg5, separable_*
    return sorted(COMMON + fam)


def kernel_source_sha(root: str, workload_name: str) -> str:
    h = hashlib.sha256()
    d = os.path.join(root, "stochastic-inventory_amd", "csrc")
    for name in kernel_files(workload_name):
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def build_source_files(root: str):
    """Everything libsdpgpu.so is built from: every file of csrc/, the public header, and build.py (its compiler flags)."""
    d = os.path.join(root, "stochastic-inventory_amd", "csrc")
    files = [os.path.join(d, f) for f in sorted(os.listdir(d)) if os.path.isfile(os.path.join(d, f))]
    return files + [os.path.join(root, "include", "sdpgpu.h"), os.path.join(root, "stochastic-inventory_amd", "build.py")]


def build_source_sha(root: str) -> str:
    """Identity of a build of libsdpgpu.so (sdpgpu_build_id returns it; build.py bakes it in): smoke() and bench.py compare
    the binary's with the tree's, so that a stale prebuilt library cannot pass for HEAD's."""
    h = hashlib.sha256()
    for path in build_source_files(root):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]
