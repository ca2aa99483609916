#!/usr/bin/env python3
"""Regenerate tests/golden/tables_*.npz.

The reference cannot run here (Java, no JDK), so these tables are NOT reference outputs: they
are the CPU oracle's value and policy tables for the small cases in tests/cases.py, frozen so
that an accidental change to the oracle (or to a case) is caught.  The only reference-recorded
numbers are in kat_reference.json.  Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import json  # noqa: E402

import cases  # noqa: E402
import multicash_cases  # noqa: E402
import staff_cases  # noqa: E402
from oracle import sdpref, staffref  # noqa: E402


def main():
    for make in cases.ALL:
        w = make()
        P = sdpref.Problem(w.desc(), w.pmf, w.overhead())
        V, pol, cells = P.solve()
        out = {"cells": np.int64(cells)}
        for t in range(w.T):
            out[f"v{t + 1}"] = V[t]
            out[f"p{t + 1}"] = pol[t].astype(np.int16)
        np.savez_compressed(os.path.join(HERE, f"tables_{w.name}.npz"), **out)
        print(w.name, cells, sum(len(v) for v in V))
    for make in staff_cases.ALL:  # STAFF family (oracle/staffref.c); the table is stored too: it comes from scipy
        c = make()
        V, pol, cells = c.oracle_problem(staffref).solve()
        out = {"cells": np.int64(cells), "table": c.table}
        for t in range(c.T):
            out[f"v{t + 1}"] = V[t]
            out[f"p{t + 1}"] = pol[t]
        np.savez_compressed(os.path.join(HERE, f"{c.name}.npz"), **out)
        print(c.name, cells, sum(len(v) for v in V))
    # CashRecursionMulti over MultiItemCash.main's parameters (oracle: literal recursion, ~2 minutes single-threaded);
    # the joint pmf is stored too: it comes from scipy
    if "--multicash" in sys.argv or not os.path.exists(os.path.join(HERE, "multicash_main.json")):
        kw = multicash_cases.main_instance()
        fv, q1, q2, states, cells = sdpref.multicash_memo(**kw)
        rec = {"final_value": fv, "q1": q1, "q2": q2, "states_per_period": states, "cells": cells,
               "pmf": [t.tolist() for t in kw["pmf"]]}
        json.dump(rec, open(os.path.join(HERE, "multicash_main.json"), "w"))
        print("multicash_main", fv, q1, q2, states, cells)
    # CashRecursionMultiXR over MultiItemCashXR.main's parameters: 6.3e10 cells; the oracle evaluates the period-2
    # states on all cores (sdpref_multixr_set_threads), same arithmetic per state as its plain recursion
    if "--multixr" in sys.argv or not os.path.exists(os.path.join(HERE, "multixr_main.json")):
        kw = multicash_cases.xr_main_instance()
        sdpref.lib().sdpref_multixr_set_threads(os.cpu_count() or 1)
        fv, y1, y2, states, cells = sdpref.multixr_memo(0.0, **kw)
        sdpref.lib().sdpref_multixr_set_threads(1)
        rec = {"final_value": fv, "y1": y1, "y2": y2, "states_per_period": states, "cells": cells,
               "pmf": [t.tolist() for t in kw["pmf"]]}
        json.dump(rec, open(os.path.join(HERE, "multixr_main.json"), "w"))
        print("multixr_main", fv, y1, y2, states, cells)


if __name__ == "__main__":
    main()
