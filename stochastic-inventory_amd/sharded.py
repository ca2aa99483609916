"""State-axis sharding over the GPUs of one node: one process per GPU.

The product path is native: `init_native_comm` + `ShardedSolver.solve_native` = sdpgpu_comm_init +
sdpgpu_solve_sharded of include/sdpgpu.h (RCCL all-gather issued by libsdpgpu.so itself; torch.distributed only
carries the 128-byte unique id).  `ShardedSolver.solve` / `solve_blocked` are the same sweep with the collective issued
through torch.distributed instead: the seam the gloo CPU tests drive, and the K-periods-per-exchange schedules.

Within a period every state is independent; between periods each rank needs the FULL V_{t+1}
(the transition can land anywhere on the grid).  So: the flat state index of every period is cut
into `world_size` equal contiguous slabs (the table row is padded to a multiple of world_size),
rank r computes V_t and the policy for slab r, then ONE all-gather of the fp64 slabs rebuilds
the full V_t on every rank, in place in the table the next period reads
(SURVEY.md section 8(e)).  Policy tables are never exchanged: they stay sharded.

xGMI is point-to-point (7 links per GPU); the message is S/world * 8 B per rank per period
(1 MB at the 1e6-state grid, 100 MB at 1e8), against seconds of compute per period, so the
exchange needs no overlap tricks -- it is issued on the compute stream right behind the kernel.

`SlabBackend` is the seam the CPU tests use: the `gloo` world_size-2 tests drive this very
class with a test double in place of the HIP engine (the product backend below has no CPU path).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from .engine import SdpEngine


def init_native_comm(engine: SdpEngine, group=None) -> bool:
    """Give `engine` (this rank's handle) its RCCL communicator INSIDE libsdpgpu.so (sdpgpu_comm_init).  rank 0's unique
    id travels over torch.distributed (any backend: gloo is enough), which is only the channel for those 128 bytes: after
    this the data path -- kernels and all-gathers -- is sdpgpu_solve_sharded.
    Returns True when EVERY rank has its communicator, False when some rank could not get one (then none keeps one): the
    ranks agree after each phase, so that a rank whose librccl does not load cannot leave the others waiting inside a
    collective."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0

    def agree(ok: bool) -> bool:
        if world == 1:
            return ok
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok), group=group)
        return all(flags)

    # phase 1, every rank, nothing collective: the device tables are allocated, the device is pinned and RCCL loads
    # (sdpgpu_comm_prepare) -- whatever can fail on ONE rank fails here, before anybody is inside ncclCommInitRank; rank 0
    # alone draws the unique id (ncclGetUniqueId starts a bootstrap thread: no reason to have one on every rank)
    uid = None
    try:
        engine.comm_prepare()
        if rank == 0:
            uid = SdpEngine.comm_unique_id()
        ok = True
    except Exception as exc:
        ok = False
        init_native_comm.last_error = str(exc)
    if not agree(ok):
        return False
    box = [uid if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    try:  # phase 2: the collective ncclCommInitRank
        engine.comm_init(box[0], rank, world)
        ok = True
    except Exception as exc:
        ok = False
        init_native_comm.last_error = str(exc)
    if not agree(ok):
        try:
            engine.comm_destroy()
        except Exception:
            pass
        return False
    return True


init_native_comm.last_error = ""


class SlabBackend:
    """What ShardedSolver needs from a per-rank engine."""

    T: int

    def slab(self, period: int):  # -> (padded, lo, hi)
        raise NotImplementedError

    def table(self, period: int) -> torch.Tensor:
        """The full padded row (this rank's copy) that must be all-gathered after run_period(period):
        8-byte elements -- the fp64 V_period row, or the engine's uint64 key row on small grids."""
        raise NotImplementedError

    def run_period(self, period: int) -> None:  # compute this rank's slab of V_period into table(period)
        raise NotImplementedError

    def finalize(self) -> None:  # enqueue any deferred read-out work (policy rows), no host wait
        pass

    def run_period_part(self, period: int, part: int) -> None:
        """part 1 = the states that need only this rank's slab of V_{period+1}, part 2 = the rest.
        Backends without a bounded dependency footprint do everything in part 2."""
        if part == 2:
            self.run_period(period)

    # -- fewer exchanges (ShardedSolver.solve_blocked) -- optional ----------------------------------------
    def footprint(self, period: int):
        """(left, right): state i of `period` reads V_{period+1}[i - left .. i + right]; None = unbounded."""
        return None

    def num_states(self, period: int) -> int:
        raise NotImplementedError

    def set_halo(self, halo: int) -> None:
        pass

    def run_period_range(self, period: int, lo: int, hi: int) -> None:
        """Compute the states [lo, hi) of `period` (a widened slab) into table(period)."""
        raise NotImplementedError


class GpuSlabBackend(SlabBackend):
    """SdpEngine on one GPU with its value arena held in a torch tensor (so RCCL can address it)."""

    def __init__(self, desc, pmf, overhead=None, device: Optional[torch.device] = None):
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        desc.device = self.device.index if self.device.index is not None else -1
        self.engine = SdpEngine(desc, pmf, overhead)
        self.T = self.engine.T
        nbytes = self.engine.values_bytes()
        self.arena = torch.zeros(nbytes // 8, dtype=torch.float64, device=self.device)
        self.engine.attach_values(self.arena.data_ptr(), nbytes)
        kbytes = self.engine.keys_bytes()
        self.key_arena = None
        if kbytes:
            self.key_arena = torch.zeros(kbytes // 8, dtype=torch.int64, device=self.device)
            self.engine.attach_keys(self.key_arena.data_ptr(), kbytes)
        self.engine.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self._views = {}

    def slab(self, period: int):
        return self.engine.slab(period)

    def table(self, period: int) -> torch.Tensor:
        """Valid once run_period(period) has been issued (which row is exchanged is decided then; the
        choice is the same in every sweep, so the view is cached)."""
        if period not in self._views:
            pad, _, _ = self.engine.slab(period)
            ptr = self.engine.exchange_ptr(period)
            for arena in (self.arena, self.key_arena):
                if arena is not None and arena.data_ptr() <= ptr < arena.data_ptr() + arena.numel() * 8:
                    base = (ptr - arena.data_ptr()) // 8
                    self._views[period] = arena[base: base + pad]
                    break
            else:
                raise RuntimeError("exchange row is not inside an arena this backend owns")
        return self._views[period]

    def table_block(self, p_lo: int, p_hi: int) -> Optional[torch.Tensor]:
        """The exchange rows of the consecutive periods p_lo..p_hi as ONE strided [rows, padded] view, when they
        lie in one arena at a constant distance (they do: the arenas are [period][padded row])."""
        rows = [self.table(p) for p in range(p_lo, p_hi + 1)]
        if len(rows) < 2:
            return rows[0].unsqueeze(0)
        pad = rows[0].numel()
        step = (rows[1].data_ptr() - rows[0].data_ptr()) // 8
        if step < pad or any(r.numel() != pad or r.dtype != rows[0].dtype for r in rows) or \
                any(rows[i + 1].data_ptr() - rows[i].data_ptr() != step * 8 for i in range(len(rows) - 1)):
            return None
        for arena in (self.arena, self.key_arena):
            if arena is not None and arena.dtype == rows[0].dtype and \
                    arena.data_ptr() <= rows[0].data_ptr() and rows[-1].data_ptr() + pad * 8 <= arena.data_ptr() + arena.numel() * 8:
                base = (rows[0].data_ptr() - arena.data_ptr()) // 8
                return arena.as_strided((len(rows), pad), (step, 1), base)
        return None

    def run_period(self, period: int) -> None:
        self.engine.run_period(period)

    def finalize(self) -> None:
        self.engine.finalize()

    def run_period_part(self, period: int, part: int) -> None:
        self.engine.run_period_part(period, part)

    def footprint(self, period: int):
        return self.engine.footprint(period)

    def num_states(self, period: int) -> int:
        return self.engine.num_states(period)

    def set_halo(self, halo: int) -> None:
        self.engine.set_halo(halo)

    def run_period_range(self, period: int, lo: int, hi: int) -> None:
        self.engine.run_period_range(period, lo, hi)

    def close(self):
        self.engine.close()


class ShardedSolver:
    """Backward sweep t = T..1 with one all-gather of V_t between periods."""

    def __init__(self, backend: SlabBackend, group=None, stage_through_host: bool = False):
        self.backend = backend
        self.group = group
        # debug only: a `gloo` group cannot address device memory, so the shard is bounced through the
        # host (used to rehearse N ranks on ONE GPU; the production path is RCCL on device memory)
        self.stage_through_host = stage_through_host
        self.force_split = False  # rehearsal/testing: take the interior + boundary path after a blocking exchange too
        self.force_exchange = False  # measurement only: issue the collective even at world_size 1 (tools/exchange_overhead.py)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.gathered_bytes = 0

    def exchange(self, period: int, async_op: bool = False):
        """All-gather the row of `period`.  With async_op the collective runs on the communicator's own
        stream (ordered after everything already queued on the compute stream) and the returned work's
        wait() makes the compute stream wait for it -- the host never blocks."""
        if self.world == 1 and not self.force_exchange:
            return None
        pad, lo, hi = self.backend.slab(period)
        full = self.backend.table(period)
        n = pad // self.world
        shard = full[self.rank * n: (self.rank + 1) * n]
        self.gathered_bytes += pad * 8
        if self.stage_through_host and full.is_cuda:
            host_full = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(host_full, shard.cpu(), group=self.group)
            full.copy_(host_full)
            return None
        # in place: this rank's shard already sits at its offset inside `full`
        return dist.all_gather_into_tensor(full, shard, group=self.group, async_op=async_op)

    def solve_native(self, overlap: bool = False, sync: bool = False) -> None:
        """The whole sweep of this rank in ONE C-ABI call: kernels and RCCL all-gathers issued by libsdpgpu.so
        (sdpgpu_solve_sharded; the communicator comes from init_native_comm).  The torch schedules below remain for
        the K-periods-per-exchange variants and for the CPU test double."""
        self.backend.engine.solve_sharded(overlap=overlap, sync=sync)

    def solve(self, first_period: int = 1, overlap: bool = True) -> None:
        """t = T..first_period.  With `overlap` (and more than one rank) the all-gather of V_{t+1} runs
        beside the INTERIOR half of period t -- the states whose cells read only this rank's slab of
        V_{t+1} -- and only the BOUNDARY half waits for it; interior + boundary = the whole period, so
        the result is the same as the sequential schedule."""
        in_flight = None  # work handle of the exchange of period+1
        for period in range(self.backend.T, first_period - 1, -1):
            if in_flight is not None:
                self.backend.run_period_part(period, 1)
                if in_flight is not True:
                    in_flight.wait()
                in_flight = None
                self.backend.run_period_part(period, 2)
            else:
                self.backend.run_period(period)
            if period > first_period:  # V_1 is never read by another period
                in_flight = self.exchange(period, async_op=overlap)
                if in_flight is None and self.force_split and self.world > 1:
                    in_flight = True
        if in_flight is not None and in_flight is not True:
            in_flight.wait()
        self.backend.finalize()  # the sweep is done when every policy row exists

    # -- K periods per exchange ---------------------------------------------------------------------------
    def plan_blocks(self, K: int, first_period: int = 1):
        """Blocks of at most K periods (descending) and, per period, how far the slab must be widened so that the
        last period of its block comes out right on the slab itself: period t of a block [t_lo .. t_hi] needs
        sum(left(tau)), sum(right(tau)) over tau = t_lo .. t-1.  None when some period has no bounded footprint."""
        T = self.backend.T
        fp = {}
        for t in range(first_period, T + 1):
            f = self.backend.footprint(t)
            if f is None:
                return None
            fp[t] = f
        blocks, halo = [], 0
        t_hi = T
        while t_hi >= first_period:
            t_lo = max(first_period, t_hi - K + 1)
            ext = {}
            for t in range(t_lo, t_hi + 1):
                ext[t] = (sum(fp[u][0] for u in range(t_lo, t)), sum(fp[u][1] for u in range(t_lo, t)))
                halo = max(halo, *ext[t])
            blocks.append((t_hi, t_lo, ext))
            t_hi = t_lo - 1
        return blocks, halo

    def prepare_blocked(self, K: int, first_period: int = 1) -> bool:
        """Call once, before the first run: sizes the backend's scratch for the widened slabs of solve_blocked(K)."""
        plan = self.plan_blocks(K, first_period)
        if plan is None:
            return False
        self.backend.set_halo(plan[1])
        return True

    def exchange_many(self, p_lo: int, p_hi: int):
        """Publish the rows of the consecutive periods p_lo..p_hi with ONE collective (a sweep of 30 us periods is
        otherwise bound by the host cost of issuing one collective per period).  Returns a completion callable:
        it makes the current stream wait for the collective and scatters the result into the rows."""
        if p_hi < p_lo:
            return None
        block = self.backend.table_block(p_lo, p_hi) if hasattr(self.backend, "table_block") else None
        if self.world == 1 and not self.force_exchange:
            return None
        if block is None or self.stage_through_host or block.shape[0] == 1:
            works = [self.exchange(p, async_op=True) for p in range(p_hi, p_lo - 1, -1)]
            return lambda: [wk.wait() for wk in works if wk is not None]
        rows, pad = block.shape
        n = pad // self.world
        self.gathered_bytes += rows * pad * 8
        inp = block[:, self.rank * n: (self.rank + 1) * n].contiguous()
        out = torch.empty((self.world * rows, n), dtype=block.dtype, device=block.device)  # rank-major concatenation
        work = dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)

        def complete():
            work.wait()
            block.view(rows, self.world, n).copy_(out.view(self.world, rows, n).permute(1, 0, 2))
        return complete

    def _blocked_program(self, K: int, first_period: int):
        key = (K, first_period)
        if getattr(self, "_programs", None) is None:
            self._programs = {}
        if key not in self._programs:
            plan = self.plan_blocks(K, first_period)
            if plan is None:
                raise RuntimeError("solve_blocked needs a bounded dependency footprint in every period")
            prog = []
            for t_hi, t_lo, ext in plan[0]:
                runs = []
                for period in range(t_hi, t_lo - 1, -1):
                    _, lo, hi = self.backend.slab(period)
                    S = self.backend.num_states(period)
                    left, right = ext[period]
                    if hi > lo:
                        runs.append((period, max(0, lo - left), min(S, hi + right)))
                    else:  # an empty slab still keeps its bookkeeping in step with the other ranks
                        runs.append((period, min(lo, S), min(lo, S)))
                prog.append((t_hi, t_lo, runs))
            self._programs[key] = prog
        return self._programs[key]

    def solve_blocked(self, K: int, first_period: int = 1) -> None:
        """The compute stream waits for an exchange only once per K periods.  Inside a block every period is
        computed on the slab widened by the footprints of the periods still to come (redundant work that
        reproduces the neighbours' values bit for bit).  Per block two collectives publish the rows: the block's
        last row in place (the next block waits for that one), the others batched (nobody waits until the sweep
        ends) -- a collective overwrites the widened part with identical bytes."""
        pending = []
        boundary = None
        for t_hi, t_lo, runs in self._blocked_program(K, first_period):
            if boundary is not None:
                boundary.wait()  # the full row of period t_hi + 1
            for period, a, b in runs:
                self.backend.run_period_range(period, a, b)
            boundary = self.exchange(t_lo, async_op=True) if t_lo > first_period else None
            done = self.exchange_many(max(t_lo + 1, first_period + 1), t_hi)
            if done is not None:
                pending.append(done)
        if boundary is not None:
            boundary.wait()
        for done in pending:
            done()
        self.backend.finalize()

    def gather_policy(self, period: int, local: torch.Tensor) -> Optional[List[torch.Tensor]]:
        """Collect the per-rank policy slabs on rank 0 (host-side read-out, not on the hot path)."""
        if self.world == 1:
            return [local]
        out = [torch.empty_like(local) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(local, out, dst=0, group=self.group)
        return out
