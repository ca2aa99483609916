"""Per-phase deadlines for a multi-rank run: fail loudly instead of hanging.

A sharded run is a sequence of phases every rank walks through together (process group, communicator, first sweep,
parity gate, schedule calibration, warm-up, timed loop, ...), most of which end in a collective.  If ONE rank stalls --
a communicator that never forms, a rank stuck before a barrier -- the others wait inside the collective for ever and the
launcher sees nothing until its own timeout, with no line and no diagnosis.  `PhaseWatchdog` gives every phase a
deadline, watched by a daemon thread of the rank itself: when a phase overruns, the rank prints ONE JSON record naming
the phase, the rank and the times (stderr, and stdout so that whoever collects the bench line finds it) and the process
ends with a non-zero code (`os._exit`: no unwinding through a blocked collective).  Nothing is relaunched or re-executed
-- a process that has touched the GPU must not be replaced -- the launcher (torchrun) then ends the remaining ranks.

    wd = PhaseWatchdog(rank, world)
    with wd.phase("communicator init", 120):
        ...
    wd.close()

SDP_WATCHDOG_SCALE multiplies every deadline (slow boxes); SDP_WATCHDOG_INJECT_STALL="<phase>:<rank>" makes that rank
sleep inside that phase (tests: a stalled rank must end the run with the record, not hang it).
"""
from __future__ import annotations

import contextlib
import json
import os
import sys
import threading
import time


class PhaseWatchdog:
    def __init__(self, rank: int, world: int, exit_code: int = 3, poll_s: float = 0.2, context: dict | None = None):
        self.rank, self.world, self.exit_code = int(rank), int(world), int(exit_code)
        self.context = dict(context or {})
        self.scale = float(os.environ.get("SDP_WATCHDOG_SCALE", "1") or 1)
        self._inject = os.environ.get("SDP_WATCHDOG_INJECT_STALL", "")
        self._trace = os.environ.get("SDP_BENCH_PHASE_TRACE", "") == "1"  # announce phases (bench.py's self-launching parent reads them)
        self._lock = threading.Lock()
        self._phase = None       # (name, t_start, deadline_s)
        self._closed = False
        self._poll = poll_s
        self.history = []        # [(name, seconds)] of the phases that finished
        self._thread = threading.Thread(target=self._watch, name="sdp-phase-watchdog", daemon=True)
        self._thread.start()

    # -- the watched side ------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def phase(self, name: str, deadline_s: float):
        limit = float(deadline_s) * self.scale
        t0 = time.monotonic()
        with self._lock:
            self._phase = (name, t0, limit)
        if self._trace:
            print(f"[bench phase] rank {self.rank}: {name}", file=sys.stderr, flush=True)
        try:
            if self._inject == f"{name}:{self.rank}":
                time.sleep(1e6)  # a stalled rank (tests)
            yield
        finally:
            with self._lock:
                self._phase = None
            self.history.append((name, time.monotonic() - t0))

    def close(self):
        self._closed = True

    def seconds(self, name: str):
        for n, s in self.history:
            if n == name:
                return s
        return None

    # -- the watching side -------------------------------------------------------------------------------------
    def _watch(self):
        while not self._closed:
            time.sleep(self._poll)
            with self._lock:
                ph = self._phase
            if ph is None:
                continue
            name, t0, limit = ph
            elapsed = time.monotonic() - t0
            if elapsed > limit:
                self._fire(name, elapsed, limit)

    def _fire(self, name: str, elapsed: float, limit: float):
        rec = {"error": "phase deadline exceeded", "phase": name, "rank": self.rank, "world": self.world,
               "elapsed_s": round(elapsed, 2), "deadline_s": round(limit, 2),
               "phases_done": [[n, round(s, 3)] for n, s in self.history], "value": None}
        rec.update(self.context)
        line = json.dumps(rec)
        try:
            # one write() per stream, record and newline together: the ranks of a job share the launcher's pipe, and on an
            # unbuffered stream print() hands over the text and the newline separately -- two ranks firing in the same instant
            # then put two records on one line (seen once in round 4: a JSONDecodeError in the reader, not a lost record)
            data = (line + "\n").encode()
            for stream in (sys.stderr, sys.stdout):
                try:
                    stream.flush()
                    os.write(stream.fileno(), data)
                except Exception:
                    print(line, file=stream, flush=True)
        finally:
            os._exit(self.exit_code)
