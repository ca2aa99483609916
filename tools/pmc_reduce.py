#!/usr/bin/env python3
"""Reduce the passes of tools/pmc_collect.sh: per kernel, the mean of every counter per launch, the rocprofv3 kernel
stats and the bench line's own HIP-event times, into
    <out>/<round>_pmc_<workload name>.json   (what bench.py's roofline.hbm / valu-issue blocks read from profiles/)
    <out>/<round>_<workload>_summary.txt     (human-readable: the table quoted in DESIGN.md)
gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports half the bytes
of a wide coalesced read and is doubled.  usage: pmc_reduce.py <out dir> <round> <workload>"""
import collections
import csv
import glob
import json
import os
import statistics
import sys


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_sha import kernel_source_sha  # noqa: E402  (same digest as bench.py's)


def short(name):
    if name.startswith("void "):  # (templated kernels carry their return type)
        name = name[len("void "):]
    if name.startswith("(anonymous namespace)::"):  # the reachable-set engine's kernels (sdpgpu_sparse.hip)
        name = "sparse::" + name[len("(anonymous namespace)::"):]
    return name.split("(")[0].replace("void ", "").strip()


def ours(k):
    return "sdp" in k or k.startswith("sparse::")


def main():
    out, rnd, wl = sys.argv[1:4]
    bench = None
    try:
        bench = json.loads([l for l in open(os.path.join(out, "bench.json")) if l.startswith("{")][-1])
    except Exception as exc:
        print("no bench line:", exc)
    counters = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> values
    prof_us = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not ours(k):
                continue
            counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    stats = {}
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if ours(k):
                stats[k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6,
                            "pct": float(r["Percentage"])}
    kernels = {}
    for k, cs in counters.items():
        rec = {c: statistics.mean(v) for c, v in cs.items()}
        rec["launches_seen"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in rec or "WRITE_SIZE" in rec:
            rec["hbm_read_bytes_per_launch"] = 2.0 * rec.get("FETCH_SIZE", 0.0) * 1024.0
            rec["hbm_write_bytes_per_launch"] = rec.get("WRITE_SIZE", 0.0) * 1024.0
        if k in stats:
            rec["rocprof"] = stats[k]
        kernels[k] = rec
    if not kernels:
        print("no counters collected")
        sys.exit(1)
    # dominant kernel: most total time under --kernel-trace --stats, else most launches
    dom = max(kernels, key=lambda k: (kernels[k].get("rocprof", {}).get("total_ms", 0.0), kernels[k]["launches_seen"]))
    d = kernels[dom]
    rec = {
        "round": rnd, "workload": bench["config"]["workload"] if bench else wl, "dominant_kernel": dom,
        "source_sha": kernel_source_sha(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                        bench["config"]["workload"] if bench else wl),
        "hbm_bytes_per_launch": d.get("hbm_read_bytes_per_launch", 0.0) + d.get("hbm_write_bytes_per_launch", 0.0),
        "valu_insts_per_launch": d.get("SQ_INSTS_VALU"),
        "ta_busy_frac": (d["TA_BUSY_avr"] / (d["GRBM_GUI_ACTIVE"] / 8.0)) if d.get("TA_BUSY_avr") and d.get("GRBM_GUI_ACTIVE") else None,
        # GRBM_GUI_ACTIVE sums the eight XCDs: GUI / 8 cycles per launch on 256 CUs x 4 SIMDs; a wave64 VALU instruction holds its
        # SIMD four cycles; SQ_LDS_IDX_ACTIVE counts LDS-array cycles per CU
        "valu_busy_frac": (d["SQ_ACTIVE_INST_VALU"] * 4.0 / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0))
        if d.get("SQ_ACTIVE_INST_VALU") and d.get("GRBM_GUI_ACTIVE") else None,
        "lds_busy_frac": (d["SQ_LDS_IDX_ACTIVE"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 256.0))
        if d.get("SQ_LDS_IDX_ACTIVE") and d.get("GRBM_GUI_ACTIVE") else None,
        # the reachable-set engine launches its kernels per period on sets of very different sizes: bench.py prices totals per
        # SOLVE, and the profiled command (pmc_collect.sh: --steps 3 --no-gate) runs the first solve + 3 timed ones
        "solves_profiled": 4 if wl.startswith("multilead") else None,
        "correction": "HBM bytes = 2 * FETCH_SIZE_KiB * 1024 + WRITE_SIZE_KiB * 1024 (gfx950: FETCH_SIZE counts half of a wide coalesced read)",
        "kernels": kernels,
        "bench_line": {k: bench[k] for k in ("value", "ms_per_step", "steps", "config", "parity_gate")} if bench else None,
        "bench_per_launch_ms_events": bench["roofline"].get("per_launch_ms_events") if bench else None,
        "command": "tools/pmc_collect.sh (separate rocprofv3 --pmc passes with --kernel-trace only; --kernel-trace --stats in its own run)",
    }
    # In-run calibration of the two traffic counters on byte counts this code knows exactly: key_fill_kernel
    # writes 8 B per state and period of the key arena -- WRITE_SIZE must read that, as is; finalize_kernel reads, per state
    # and period, its 8-byte key plus the (value, action) rows it scans (12 B per row that decides) -- with FETCH_SIZE
    # DOUBLED that comes out at a whole number of rows (1.0005 on the target grid), un-doubled it would be half a key short.
    if bench and "sdp::key_fill_kernel" in kernels and "WRITE_SIZE" in kernels["sdp::key_fill_kernel"]:
        cfg = bench["config"]
        st = float(cfg["states"]) * float(cfg["periods"])
        cal = {"key_fill_write_ratio": kernels["sdp::key_fill_kernel"]["WRITE_SIZE"] * 1024.0 / (8.0 * st)}
        fk = kernels.get("sdp::finalize_kernel")
        if fk and "FETCH_SIZE" in fk:
            cal["finalize_rows_read_per_state"] = (2.0 * fk["FETCH_SIZE"] * 1024.0 - 8.0 * st) / (12.0 * st)
        rec["calibration"] = cal
        if not (0.98 <= cal["key_fill_write_ratio"] <= 1.02):
            print("WRITE_SIZE calibration failed:", cal)
            sys.exit(2)
        if "finalize_rows_read_per_state" in cal and cal["finalize_rows_read_per_state"] < 0.9:
            print("FETCH_SIZE calibration failed (the doubling does not hold):", cal)
            sys.exit(2)
    name = rec["workload"]
    json.dump(rec, open(os.path.join(out, f"{rnd}_pmc_{name}.json"), "w"), indent=1)
    lines = [f"{rnd} {name}: dominant kernel {dom}"]
    if bench:
        rf = bench["roofline"]
        if "per_launch_ms_events" in rf:
            lines.append(f"bench line: {bench['value']:.4g} cells/s, {bench['ms_per_step']:.4f} ms per sweep, avg launch {rf['avg_launch_ms'] * 1e3:.2f} us (HIP events, un-profiled); "
                         f"sum of per-launch events {sum(rf['per_launch_ms_events']):.4f} ms; roofline {rf['bound']} frac {rf['frac']}")
        else:  # (the reachable-set engine: a step is a whole solve)
            lines.append(f"bench line: {bench['value']:.4g} cells/s, {bench['ms_per_step']:.4f} ms per solve (wall), device "
                         f"{rf.get('device_ms_per_solve')} ms per solve (HIP events); roofline {rf['bound']} frac {rf['frac']}")
    for k, r in sorted(kernels.items(), key=lambda kv: -kv[1].get("rocprof", {}).get("total_ms", 0.0)):
        rp = r.get("rocprof")
        lines.append(f"-- {k}" + (f": {rp['calls']} calls, avg {rp['avg_us']:.2f} us under rocprofv3 ({rp['pct']:.1f} % of kernel time)" if rp else ""))
        for c in sorted(r):
            if c in ("rocprof", "launches_seen"):
                continue
            lines.append(f"     {c}: {r[c]:.6g}")
        if r.get("SQ_INSTS_VALU") and r.get("SQ_BUSY_CYCLES"):
            pass
    open(os.path.join(out, f"{rnd}_{wl}_summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
