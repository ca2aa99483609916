#!/usr/bin/env python3
"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of
`python bench.py` into profiles/pmc_traffic.json, the per-launch HBM traffic bench.py reports.

gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports exactly
half the bytes of a wide coalesced read, so it is doubled.  The doubling is calibrated in-run on a
known byte count in this code's own access pattern: the combine kernel reads n_chunks * S * 12 B of
partial rows and FETCH_SIZE shows half of that.

usage: tools/collect_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <bench.json> [round]
"""
import csv
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, counter):
    by = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            by.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    return by


def main():
    fetch, write, bench = sys.argv[1:4]
    rnd = sys.argv[4] if len(sys.argv) > 4 else "r01"
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    b = json.loads(open(bench).read().strip().splitlines()[-1])
    kernels = {}
    for k in sorted(set(f) | set(w)):
        if "sdp::" not in k:
            continue
        fk, wk = f.get(k, [0.0]), w.get(k, [0.0])
        kernels[k] = {
            "launches": len(fk),
            "FETCH_SIZE_KiB_mean": statistics.mean(fk),
            "WRITE_SIZE_KiB_mean": statistics.mean(wk),
            "hbm_read_bytes_per_launch": 2.0 * statistics.mean(fk) * 1024.0,
            "hbm_write_bytes_per_launch": statistics.mean(wk) * 1024.0,
        }
    # dominant kernel = the one with a future term and the most launches
    dom = max((k for k in kernels if "combine" not in k and "reach" not in k), key=lambda k: kernels[k]["launches"])
    rec = {
        "round": rnd,
        "workload": b["config"]["workload"],
        "kernel_used": {"auto": 0, "gather": 1, "window": 2}[b["config"]["kernel"]],
        "dominant_kernel": dom,
        "hbm_bytes_per_launch": kernels[dom]["hbm_read_bytes_per_launch"] + kernels[dom]["hbm_write_bytes_per_launch"],
        "correction": "bytes = 2 * FETCH_SIZE_KiB * 1024 + WRITE_SIZE_KiB * 1024 (gfx950: FETCH_SIZE counts half of a coalesced read)",
        "kernels": kernels,
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline",
    }
    out = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    json.dump(rec, open(out, "w"), indent=1)
    json.dump(rec, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_{rec['workload']}.json"), "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
