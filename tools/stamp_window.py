#!/usr/bin/env python3
"""Diagnostic only: run one cfg2 sweep on the SDP_STAMPS build (tools/libsdpgpu_stamps.so) and print the
per-wave timeline of one mid-sweep window-kernel launch: when waves start, how long staging and the
demand loop take, how many waves each SIMD hosted.  Never part of the product or of any timing."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
sia._abi.LIB_PATH = os.path.join(ROOT, "tools", "libsdpgpu_stamps.so")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
os.chdir(ROOT)
w = workloads.cfg2_clsp(T=6)
eng = sia.SdpEngine(w.desc(), w.pmf)
eng.solve(); eng.solve()
rows = [list(map(int, l.split())) for l in open("gpurun_out/stamps.txt")]
rows = [r for r in rows if r[1] and r[3]]
t0 = min(r[1] for r in rows)
ends = [(r[3] - t0) / 100.0 for r in rows]       # memrealtime: 100 MHz -> us
starts = [(r[1] - t0) / 100.0 for r in rows]
stage = [(r[2] - r[1]) / 100.0 for r in rows]
loop = [(r[3] - r[2]) / 100.0 for r in rows]
import statistics as st
print(f"waves {len(rows)}  kernel span {max(ends):.2f} us")
print(f"start: median {st.median(starts):.2f} p90 {sorted(starts)[int(.9*len(starts))]:.2f} max {max(starts):.2f}")
print(f"staging: median {st.median(stage):.2f} max {max(stage):.2f};  loop: median {st.median(loop):.2f} min {min(loop):.2f} max {max(loop):.2f}")
per = collections.Counter()
for r in rows:
    hw, xcc = r[4], r[5]
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    per[(xcc & 15, se, sh, cu, simd)] += 1
c = collections.Counter(per.values())
print("SIMDs used", len(per), "waves-per-SIMD histogram", sorted(c.items()))
hist = collections.Counter(int(e) for e in ends)
print("end-time histogram (us: waves):", sorted(hist.items()))
hs = collections.Counter(int(e) for e in starts)
print("start-time histogram (us: waves):", sorted(hs.items()))
