"""A chromosome block resident in HBM and its repeated scoring (the path ``score``, bench.py and the
multi-GPU driver all run): buffers are allocated once, one ``step()`` enqueues the whole hot path --
site_counts (+ fused per-site decision) -> window_bounds -> window_stats -> async copy of the records
to pinned host memory -- without any host synchronisation: the genotype stream on the current HIP stream,
what follows it on the scorer's second stream, behind a stream-side wait (``results()`` waits for both).

A block may hold several chromosome *pieces* back to back (a rank's share of a whole-genome window
list, sai_amd.sharding): positions then ascend only inside a piece, every window names the piece
("segment") it is searched in, and one site pass + one windows stage still cover the whole block.
"""

from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _ffi
from .engine import PLANES, RECORD_DTYPE, Engine, WindowResults
from .utils.windows import split_genome

# A step is milliseconds, but in a multi-GPU job a windows stage ends with a gather, and the first
# gather of a job waits for RCCL to build its channels (seconds on 8 ranks): the deadline only has
# to beat "forever".  SAI_AMD_WAIT_DEADLINE_S overrides it.
# How the host waits for an event of the pipelined scorer: "sync" = Event.synchronize() (the runtime
# wakes the thread 8-12 us after the kernel ends, tools/event_wait_probe.py, and no core spins);
# "poll" = busy-poll Event.query() with a deadline that raises instead of waiting forever -- for
# hunting a hang.  (Round 1 polled because synchronize() seemed to oversleep by milliseconds; that
# belonged to the stream-side-wait arrangement it had at the time and does not reproduce.)
_WAIT_MODE = __import__("os").environ.get("SAI_AMD_WAIT", "sync")
WAIT_DEADLINE_S = float(__import__("os").environ.get("SAI_AMD_WAIT_DEADLINE_S", "300"))
_JOIN_COPY = __import__("os").environ.get("SAI_AMD_JOIN_COPY", "1") != "0"  # 0: records and list heads as three copies (A/B runs)


@dataclass
class ResidentBlock:
    """ref, tgt and source populations of one contiguous site range (or of several pieces laid
    end to end at tile boundaries), plus positions."""

    pops: list  # [ref, tgt, src...] TiledPop
    ploidies: list  # same order
    pos: "object"  # int32 device tensor [n_sites]
    # block-relative [lo, hi) site range of every piece; None = the block is one ascending run
    segments: Optional[list] = None
    extra: dict = field(default_factory=dict)  # e.g. the outgroup block of the ABBA-BABA family

    @property
    def n_sites(self) -> int:
        return self.pops[0].n_sites

    @property
    def n_real_sites(self) -> int:
        """Sites that carry data (pieces are padded to whole tiles in between)."""
        if self.segments is None:
            return self.n_sites
        return sum(hi - lo for lo, hi in self.segments)

    @property
    def genotype_bytes(self) -> int:
        """Algorithmic bytes of one site_counts launch: every genotype byte once."""
        return self.n_real_sites * sum(p.n_ind for p in self.pops)

    @property
    def packed2_bytes(self) -> int:
        """Algorithmic bytes of one packed2 launch: two bits per genotype, each site's row rounded
        up to whole bytes (the layout's padding of every site to 64-individual groups is not counted)."""
        return self.n_real_sites * sum((p.n_ind + 3) // 4 for p in self.pops)


def synth_block(eng: Engine, seed: int, chrom: int, n_sites: int, n_ref: int, n_tgt: int, n_src_list: Sequence[int],
                ploidy: int = 2, missing_per_million: int = 0, site0: int = 0) -> ResidentBlock:  # fmt: skip
    """synth-v1 block generated in place on the GPU."""
    sizes = [n_ref, n_tgt] + list(n_src_list)
    pops = [
        eng.synth_population(seed, chrom, site0, n_sites, stream, n, ploidy, missing_per_million)
        for stream, n in enumerate(sizes)
    ]
    pos = eng.synth_positions(seed, chrom, n_sites, site0=site0)
    return ResidentBlock(pops, [ploidy] * len(sizes), pos)


class _SetChunk:
    """Window-stage buffers of up to SAI_MAX_SETS parameter sets (one sai_window_stats call)."""

    def __init__(self, eng: Engine, s0: int, s1: int, n_windows: int, cap_u: int, cap_q: int, fetch_lists: int = 0):
        import torch

        self.s0, self.s1 = s0, s1
        # the whole of both lists is fetched with the records (the product path: lists no longer than they are
        # fetched): records and lists lie in one allocation and travel as ONE copy
        self.joined = _JOIN_COPY and 0 < max(int(cap_u), 1) <= int(fetch_lists) and 0 < max(int(cap_q), 1) <= int(fetch_lists)
        self.bufs = eng.alloc_window_bufs(s1 - s0, n_windows, cap_u, cap_q, joined=self.joined)
        self.dev_whole = self.bufs[6] if self.joined else None
        head_bytes = self.bufs[5].numel()
        # pinned mirror of the records | offsets | totals buffer: one copy per step
        self._eng, self._pinned = eng, eng.pinned_acquire(self.dev_whole.numel() if self.joined else head_bytes)
        # ... and, when asked for, of the first `fetch_lists` entries of both candidate lists: their sizes
        # are known only from the totals, so the lists would otherwise cost a second round trip
        self.n_fetch = (min(int(fetch_lists), self.bufs[2].numel()), min(int(fetch_lists), self.bufs[3].numel()))
        self._pinned_lists = eng.pinned_acquire(4 * sum(self.n_fetch)) if sum(self.n_fetch) and not self.joined else None
        self.host_lists = self.host_whole = None
        if self.joined:
            self.host_whole = self._pinned[: self.dev_whole.numel()]
            flat = self.host_whole[head_bytes:].view(torch.int32)
            self.host_lists = (flat[: self.n_fetch[0]], flat[self.n_fetch[0] :])
        elif self._pinned_lists is not None:
            flat = self._pinned_lists[: 4 * sum(self.n_fetch)].view(torch.int32)
            self.host_lists = (flat[: self.n_fetch[0]], flat[self.n_fetch[0] :])
        self.host_head = self._pinned[:head_bytes]
        rec_bytes = (s1 - s0) * n_windows * RECORD_DTYPE.itemsize
        self.rec_bytes = rec_bytes
        self.host_records = self.host_head[:rec_bytes]
        off_bytes = (s1 - s0) * n_windows * 16
        self.host_offsets = self.host_head[rec_bytes : rec_bytes + off_bytes].view(torch.int64)
        self.host_totals = self.host_head[rec_bytes + off_bytes : rec_bytes + off_bytes + 16].view(torch.int64)

    def release(self, streams=()) -> None:
        """Hand the pinned mirror back to the engine's pool (the chunk must not be used afterwards)."""
        if self._pinned is not None:
            buf, self._pinned = self._pinned, None
            self.host_head = self.host_records = self.host_offsets = self.host_totals = self.host_whole = None
            if self.joined:
                self.host_lists = None
            self._eng.pinned_release(buf, streams)
        if self._pinned_lists is not None:
            buf, self._pinned_lists, self.host_lists = self._pinned_lists, None, None
            self._eng.pinned_release(buf, streams)


class ResidentScorer:
    def __init__(self, eng: Engine, block: ResidentBlock, windows: Sequence[tuple], sets: Sequence[_ffi.SaiParams],
                 cap_u: int = 1 << 20, cap_q: int = 1 << 20, layout: str = "int8", overlap: bool = False,
                 window_segment: Optional[Sequence[int]] = None, counts_out=None, counts_in=None,
                 lists_as_indices: bool = False, fetch_lists: int = 0, dd_out=None):  # fmt: skip
        """``windows`` = inclusive ``(start, end)`` position pairs; for a block of several pieces
        ``window_segment[w]`` is the index into ``block.segments`` of the piece window w lies in.

        ``layout="packed2"`` re-encodes the block once into the 2-bit layout (dosages 0..2 only)
        and streams that instead: 4x fewer genotype bytes per step, identical results.

        ``overlap=True`` software-pipelines consecutive steps: the windows stage of step k
        (bounds, statistics, candidate lists, copy to the host) runs on a second HIP stream while
        the site pass of step k+1 already streams genotypes on the caller's stream; the per-site
        arrays are double-buffered and events order every reuse.  Same kernels, same results.

        ``counts_out`` (int32 device tensor [P][n_sites][2]) additionally receives the per-population
        ``{alt_sum, n_called}`` of the pass (what the ABBA-BABA family divides).  ``counts_in`` (same
        shape) says the counts of these populations already exist -- several population combinations
        share blocks that were reduced once -- so a step starts at the per-site decision and no
        genotype byte is read.  ``lists_as_indices``: the candidate lists hold block-relative
        site indices instead of positions.  ``fetch_lists`` = n: every step also copies the first n
        entries of both candidate lists to pinned host memory with the records, and ``results()`` needs
        no second round trip when the lists are that short (the product path; bench.py's steps gather rows).
        ``dd_out`` = (index of the first source population, number of them, int32 device tensor [2][rows][n_sites]):
        DD's per-site terms of those populations' individuals ride along the fused pass (``Engine.site_pass_dd``;
        the caller has asked ``Engine.dd_rides_along``)."""
        import torch

        self.fetch_lists = int(fetch_lists)
        if layout not in ("int8", "packed2"):
            raise ValueError("layout must be 'int8' or 'packed2'")
        self.layout = layout
        self.packed = [eng.pack2(p) for p in block.pops] if layout == "packed2" else None

        if len(sets) < 1:
            raise ValueError("at least one parameter set per scorer")
        self.eng, self.block, self.sets = eng, block, list(sets)
        # (start, end) pairs or an int64 array [n_windows][2] (no Python object per window)
        self.windows = np.ascontiguousarray(np.asarray(windows, dtype=np.int64).reshape(-1, 2).T)  # [2][n_windows]
        n, n_w, n_s = block.n_sites, int(self.windows.shape[1]), len(self.sets)
        self.n_windows, self.n_sets = n_w, n_s
        dev = eng.device
        both = torch.as_tensor(self.windows).to(dev)
        self.win_start, self.win_end = both[0], both[1]
        self.seg_lo = self.seg_hi = None
        if block.segments is not None:
            if window_segment is None:
                if len(block.segments) != 1:
                    raise ValueError("a block of several pieces needs window_segment")
                window_segment = np.zeros(n_w, dtype=np.int64)
            seg = np.asarray(block.segments, dtype=np.int64).reshape(-1, 2)
            ws = np.asarray(window_segment, dtype=np.int64)
            if ws.shape != (n_w,) or (n_w and (ws.min() < 0 or ws.max() >= len(seg))):
                raise ValueError("window_segment does not match the block's segments")
            if len(seg) and (seg[:, 0].min() < 0 or seg[:, 1].max() > n or np.any(seg[:, 1] < seg[:, 0])):
                raise ValueError("segment outside the block")
            self.seg_lo = torch.as_tensor(seg[ws, 0].astype(np.int32)).to(dev)
            self.seg_hi = torch.as_tensor(seg[ws, 1].astype(np.int32)).to(dev)
        elif window_segment is not None:
            raise ValueError("window_segment given for a block without segments")
        # at most SAI_FUSED_SETS sets: one fused launch, the per-population counts never leave the chip
        if counts_in is not None and (counts_out is not None or layout != "int8"):
            raise ValueError("counts_in excludes counts_out and the packed2 layout")
        self.have_counts = counts_in is not None
        # one fused launch serves at most SAI_FUSED_SETS sets and SAI_FUSED_SRC source populations; beyond that the
        # counts (in groups of populations) and the stand-alone per-site decision
        self.fused = n_s <= _ffi.SAI_FUSED_SETS and not self.have_counts and len(block.pops) <= 2 + _ffi.SAI_FUSED_SRC
        if dd_out is not None and (not self.fused or layout != "int8"):
            raise ValueError("DD rides along the fused int8 pass only")
        self.dd_out = dd_out
        self.counts_out = counts_out
        self.counts = counts_in if self.have_counts else counts_out
        if self.counts is None and not self.fused:
            self.counts = torch.empty((len(block.pops), n, 2), dtype=torch.int32, device=dev)
        self.list_pos = None if lists_as_indices else block.pos
        self.overlap = bool(overlap)
        # three sets: the windows stage of step k runs under site pass k+1, and the host may enqueue site
        # pass k+2 without waiting for it (with two sets a short pass -- C2: 75 us -- left the main
        # stream idle whenever the stage under it ran longer than the pass)
        n_buf = 3 if self.overlap else 1
        # the fused pass writes tgt_freq only at candidate sites (all the windows stage reads).  What the pass writes lies
        # best in memory of another class than the populations it reads (sai_amd/placement.py): in the block's output
        # arena when the placement search found such memory and this scorer still fits into it
        arena = (block.extra or {}).get("output_arena")
        n_tiles = (n + _ffi.SAI_TILE_SITES - 1) // _ffi.SAI_TILE_SITES

        def in_arena(nbytes: int):
            return arena.take(nbytes) if arena is not None and nbytes > 0 else None

        self._tgt_freq, self._flags = [], []
        for _ in range(n_buf):
            freq, planes = in_arena(8 * n), in_arena(8 * n_tiles * PLANES * n_s)
            if freq is None or planes is None:
                arena = None  # used up: this and the remaining sets as always
                freq, planes = torch.full((n,), float("nan"), dtype=torch.float64, device=dev), eng.alloc_planes(n, n_s)
            else:
                freq = freq.view(torch.float64).fill_(float("nan"))
                planes = planes.view(torch.int64).reshape(n_tiles, PLANES * n_s).zero_()
            self._tgt_freq.append(freq)
            self._flags.append(planes)  # flag planes [tiles][3 * sets]
        # the windows stage always runs on a second stream.  Pipelined form: under the next site pass.  Plain
        # form: a kernel queued BEHIND a running site pass on the pass's own stream slows the pass down (C3, one
        # box: 2.92 ms with nothing or only a copy behind it, 3.09 with window_bounds, 3.10 with the whole stage;
        # profiles/history/r04_lone_pass.txt), so the stage waits for the pass on its own stream instead: a lone
        # step + results takes 3.02 instead of 3.19 ms
        self.side = torch.cuda.Stream(device=dev, priority=-1)  # small kernels first
        self._plain_done = [torch.cuda.Event() for _ in range(n_buf)]
        self._site_done = list(self._plain_done)  # per buffer set: what the host waits for before the windows stage
        # the hand-over (and, in timed steps, both ends) of a fused pass ride in its own dispatch packet
        self._carried = [None] * n_buf  # per buffer set: the LaunchEvent pair its plan carries at the moment
        self._win_done = [torch.cuda.Event() for _ in range(n_buf)]  # after the windows stage that last read buffer b
        self._win_used = [False] * n_buf
        self._pending = None  # (buffer set, step index) whose windows stage has not been enqueued yet
        self.after_stage = None  # optional callable(step_index), run on the window stream right after a stage
        self._k = 0
        self.lo = torch.empty((n_w,), dtype=torch.int32, device=dev)
        self.hi = torch.empty((n_w,), dtype=torch.int32, device=dev)
        self._alloc_chunks(cap_u, cap_q)
        self._build_pass_plans()
        # timed site passes (``step(time_counts=True)``): their durations in ms.  The event pairs behind them are a
        # small ring per buffer set -- a pair is read and reused once its pass has certainly ended -- not a list that
        # grows with every timed step; ``close()`` releases them
        self._pass_ms: list = []
        self._timing: list = [[] for _ in range(n_buf)]  # per buffer set: [start, stop, in_flight] entries

    def _alloc_chunks(self, cap_u: int, cap_q: int) -> None:
        m = _ffi.SAI_MAX_SETS
        self.cap_u, self.cap_q = int(cap_u), int(cap_q)
        self._release_chunks()
        self.chunks = [
            _SetChunk(self.eng, s0, min(s0 + m, self.n_sets), self.n_windows, cap_u, cap_q, self.fetch_lists)
            for s0 in range(0, self.n_sets, m)
        ]
        first = self.chunks[0]  # the whole scorer when n_sets <= SAI_MAX_SETS
        self.bufs, self.host_head = first.bufs, first.host_head
        self.host_records, self.host_offsets, self.host_totals = first.host_records, first.host_offsets, first.host_totals
        self._build_stage_plans()

    def rebind(self, block: ResidentBlock, sets: Sequence[_ffi.SaiParams], counts_in=None,
               lists_as_indices: bool = False, dd_out=None) -> None:
        """Point this scorer at another block over the SAME sites and windows -- the next population
        combination of a region, or the same one again -- and other parameter sets of the same number:
        the per-site arrays, the window arrays on the device, the candidate buffers and their pinned
        mirrors stay; only the prepared launch sequences are recorded again.  (A scorer per combination
        cost about a millisecond to build for a pass of three: the NaN fill of 80 MB of tgt_freq, the
        upload of the windows, the allocations.)"""
        if self.layout != "int8" or self.counts_out is not None:
            raise ValueError("rebind serves the plain int8 scorer")
        if block.n_sites != self.block.n_sites or len(block.pops) < 2 or len(sets) != self.n_sets:
            raise ValueError("rebind needs a block of the same number of sites and as many parameter sets")
        if (block.segments is None) != (self.block.segments is None) or (block.segments is not None and block.segments != self.block.segments):
            raise ValueError("rebind needs the same chromosome pieces")
        if dd_out is not None and (counts_in is not None or self.n_sets > _ffi.SAI_FUSED_SETS or len(block.pops) > 2 + _ffi.SAI_FUSED_SRC):
            raise ValueError("DD rides along the fused int8 pass only")
        signature = self._binding_signature(block, sets, counts_in, lists_as_indices, dd_out)
        if signature == getattr(self, "_bound", None):
            # the very same buffers and parameters as the launch sequences in hand were recorded for (the same
            # region scored again): nothing to wait for, nothing to record
            self.block = block
            return
        self._sync()  # nothing of the old block may still be in flight
        self._bound = signature
        self.block, self.sets = block, list(sets)
        self.have_counts = counts_in is not None
        self.fused = self.n_sets <= _ffi.SAI_FUSED_SETS and not self.have_counts and len(block.pops) <= 2 + _ffi.SAI_FUSED_SRC
        self.dd_out = dd_out
        self.counts = counts_in
        if self.counts is None and not self.fused:
            import torch

            self.counts = torch.empty((len(block.pops), block.n_sites, 2), dtype=torch.int32, device=self.eng.device)
        self.list_pos = None if lists_as_indices else block.pos
        self._build_pass_plans()
        self._build_stage_plans()

    @staticmethod
    def _binding_signature(block, sets, counts_in, lists_as_indices, dd_out=None):
        """What the prepared launch sequences depend on: where the blocks and positions lie, the ploidies, the
        parameter sets byte for byte, the counts handed in, positions or indices in the lists, where DD's terms go."""
        import ctypes as C

        return (
            tuple((p.tiles.data_ptr(), p.n_sites, p.n_ind) for p in block.pops), tuple(block.ploidies), block.pos.data_ptr(),
            tuple(bytes(C.string_at(C.addressof(ps), C.sizeof(ps))) for ps in sets),
            None if counts_in is None else counts_in.data_ptr(), bool(lists_as_indices),
            None if dd_out is None else (dd_out[0], dd_out[1], dd_out[2].data_ptr()),
        )  # fmt: skip

    # A step's launches are recorded once per buffer set as prepared sequences (Engine.plan): the
    # host then pays for two C calls per step instead of nine with freshly marshalled arguments.
    def _build_pass_plans(self) -> None:
        eng, blk = self.eng, self.block
        self._carried = [None] * len(self._tgt_freq)  # new plans carry nothing yet
        self._count_plans, self._flag_plans = [], []
        for tgt_freq, planes in zip(self._tgt_freq, self._flags):
            out = (tgt_freq, planes)
            cp = fp = None
            if self.packed is not None:
                cp = eng.plan()
                if self.fused:
                    cp.add_site_pass(self.packed, blk.ploidies, self.sets, out, counts=self.counts_out, freq_mode="candidates",
                                     packed2=True)  # fmt: skip
                else:
                    cp.add_site_pass(self.packed, blk.ploidies, [], None, counts=self.counts, packed2=True)
            elif self.fused:
                cp = eng.plan()
                cp.add_site_pass(blk.pops, blk.ploidies, self.sets, out, counts=self.counts_out, freq_mode="candidates",
                                 dd=self.dd_out)  # fmt: skip
            elif not self.have_counts:
                cp = eng.plan()
                cp.add_site_counts(blk.pops, self.counts)
            if not self.fused:
                fp = eng.plan()
                fp.add_site_flags(self.counts, blk.ploidies, self.sets, out)
            self._count_plans.append(cp)
            self._flag_plans.append(fp)

    def _build_stage_plans(self) -> None:
        eng, blk = self.eng, self.block
        self._stage_plans = []
        for tgt_freq, planes in zip(self._tgt_freq, self._flags):
            sp = eng.plan()
            sp.add_window_bounds(blk.pos, self.win_start, self.win_end, self.seg_lo, self.seg_hi, self.lo, self.hi)
            for ch in self.chunks:
                sp.add_window_stats(tgt_freq, planes[:, PLANES * ch.s0 : PLANES * ch.s1], self.sets[ch.s0 : ch.s1], self.lo,
                                    self.hi, self.list_pos, ch.bufs)  # fmt: skip
                if ch.joined:
                    sp.add_copy_to_host(ch.host_whole, ch.dev_whole)
                    continue
                sp.add_copy_to_host(ch.host_head, ch.bufs[5])
                if ch.host_lists is not None:
                    for host, dev, n in zip(ch.host_lists, ch.bufs[2:4], ch.n_fetch):
                        if n:
                            sp.add_copy_to_host(host, dev[:n])
            self._stage_plans.append(sp)

    def _release_chunks(self) -> None:
        for ch in getattr(self, "chunks", None) or ():
            ch.release((self.side,) if getattr(self, "side", None) is not None else ())
        self.chunks = []

    def site_pass_ms(self) -> list:
        """Durations (ms) of the timed site passes so far, in step order; waits for the ones still in flight."""
        for ring in self._timing:
            for pair in ring:
                if pair[2]:
                    pair[1].synchronize()
        self._drain_timing()
        return list(self._pass_ms)

    def _drain_timing(self, b=None) -> None:
        rings = self._timing if b is None else [self._timing[b]]
        done = []
        for ring in rings:
            for pair in ring:
                if pair[2] and pair[1].query():
                    done.append((pair[2], pair[0].elapsed_time(pair[1])))
                    pair[2] = 0
        self._pass_ms.extend(ms for _, ms in sorted(done))

    def _timing_pair(self, b: int, make):
        """A free (start, stop) pair of buffer set b's ring, marked in flight with the step's ordinal."""
        self._drain_timing(b)
        ring = self._timing[b]
        pair = next((p for p in ring if not p[2]), None)
        if pair is None:
            pair = [make(), make(), 0]
            ring.append(pair)
        pair[2] = self._k + 1
        return pair

    def close(self) -> None:
        """Return the pinned host mirrors to the engine and release the timing events (also done when the scorer
        is collected)."""
        self._release_chunks()
        self._timing = [[] for _ in self._timing]
        self._carried = [None] * len(self._carried)

    def __del__(self):
        try:
            self._release_chunks()
        except Exception:  # interpreter shutdown: torch may be half gone
            pass

    @property
    def tgt_freq(self):
        """Per-site buffers of the most recent step."""
        return self._tgt_freq[(self._k - 1) % len(self._tgt_freq)]

    @property
    def flags(self):
        """Flag planes of the most recent step (Engine.alloc_planes' layout)."""
        return self._flags[(self._k - 1) % len(self._flags)]

    def flag_bytes(self):
        """The most recent step's decisions as one byte per set and site (Engine.flag_bytes)."""
        self._sync()  # the per-site arrays may have been written on the second stream (site_flags of a pass that is not the fused one)
        return self.eng.flag_bytes(self.flags, self.block.n_sites, self.sets, self.tgt_freq)

    def site_tgt_freq(self):
        """The most recent step's target frequencies per site (Engine.site_tgt_freq): NaN where none was stored."""
        self._sync()
        return self.eng.site_tgt_freq(self.flags, self.tgt_freq, self.block.n_sites)

    def window_stream(self):
        """Context manager selecting the stream on which window records are produced (for follow-up
        work such as the multi-GPU gather of the records).  In the pipelined form call ``flush()``
        first, or use ``after_stage``: the newest step's stage is enqueued one step late."""
        import contextlib

        import torch

        return torch.cuda.stream(self.side)

    def _wait(self, event, what: str) -> None:
        """Wait for an event on the host (see _WAIT_MODE).  The polling form spins for the first
        millisecond, then yields the core between polls; it raises instead of spinning forever when
        the GPU has stopped making progress, and surfaces a failed kernel's status (query() raises)."""
        if _WAIT_MODE == "sync":
            event.synchronize()
            return
        t0 = time.perf_counter()
        spins = 0
        while not event.query():
            spins += 1
            if spins & 0xFF:
                continue
            dt = time.perf_counter() - t0
            if dt > WAIT_DEADLINE_S:
                raise RuntimeError(
                    f"ResidentScorer: {what} of step {self._k} did not finish within {WAIT_DEADLINE_S:.0f} s "
                    f"(buffer set {self._k % len(self._flags)}); the GPU is hung or a kernel faulted"
                )
            if dt > 1e-3:
                time.sleep(0)

    def _launch_pending(self) -> None:
        """Pipelined form: enqueue the windows stage of the pending step on the second stream, as soon
        as the host sees its site pass finished.  Host-driven on purpose -- a stream-side wait makes
        the second queue sit behind a cross-queue barrier, and right after a synchronisation such a
        queue was not scheduled until the main queue drained (a 1.5-10 ms bubble per burst of steps),
        while dozens of queued steps made every other site pass trail its dependency by 1-2 ms."""
        import torch

        if self._pending is None:
            return
        b, index = self._pending
        self._pending = None
        self._wait(self._site_done[b], "site pass")
        with torch.cuda.stream(self.side):
            self._stage_plans[b].run()
            if self.after_stage is not None:
                self.after_stage(index)
            self._win_done[b].record(self.side)
            self._win_used[b] = True
        self.side.query()  # submit now: the runtime batches a stream's commands until something asks

    def flush(self) -> None:
        """Enqueue whatever the pipelined form still holds back (no-op otherwise)."""
        if self.overlap:
            self._launch_pending()

    def step(self, time_counts: bool = False) -> None:
        import torch

        eng = self.eng
        b = self._k % len(self._flags)
        main = torch.cuda.current_stream(eng.device)
        if not self.overlap and self._win_used[b]:
            main.wait_event(self._win_done[b])  # the stage of the step before reads the one buffer set this pass writes
        if self.overlap and self._win_used[b]:
            self._wait(self._win_done[b], "windows stage")  # the stage that last read buffer set b (2 steps ago)
        # Pipelined form, fused pass: the events ride in the pass's own dispatch packet (saihip.h,
        # sai_plan_set_pass_events) -- between two consecutive passes of the queue there is then NO marker packet
        # (end of pass k, hand-over, start of pass k + 1 were three: ~10 us of a 75-us pass).  A timed step gets a
        # pair of its own (its duration is read after the run); an untimed one reuses its buffer set's pair.
        carried = self.overlap and self._flag_plans[b] is None and self._count_plans[b] is not None
        if carried:
            from .engine import LaunchEvent

            if time_counts:  # a pair of the ring: its pass of three steps ago (the same buffer set) has ended
                slot = self._timing_pair(b, lambda: LaunchEvent(eng))
                pair = (slot[0], slot[1], True)
                self._count_plans[b].set_pass_events(pair[0], pair[1])
                self._carried[b] = pair
            elif self._carried[b] is None or self._carried[b][2]:
                pair = (LaunchEvent(eng), LaunchEvent(eng), False)
                self._count_plans[b].set_pass_events(pair[0], pair[1])
                self._carried[b] = pair
            pair = self._carried[b]
            self._count_plans[b].run()
            self._site_done[b] = pair[1]
        else:
            if time_counts:
                e0, e1, _ = self._timing_pair(b, lambda: torch.cuda.Event(enable_timing=True))
                e0.record()
            if self._count_plans[b] is not None:
                self._count_plans[b].run()  # the genotype stream: site_counts, fused with the per-site decision when it can be
            if time_counts:
                e1.record()
            if self._flag_plans[b] is not None and self.overlap:
                self._flag_plans[b].run()
        index = self._k
        self._k += 1
        if not self.overlap:
            # plain form: whatever follows the genotype stream is enqueued on the second stream, behind a
            # stream-side wait for the pass (nothing but markers stands behind the pass on its own stream)
            self._plain_done[b].record(main)
            self.side.wait_event(self._plain_done[b])
            with torch.cuda.stream(self.side):
                if self._flag_plans[b] is not None:
                    self._flag_plans[b].run()
                self._stage_plans[b].run()
                if self.after_stage is not None:
                    self.after_stage(index)
                self._win_done[b].record(self.side)
                self._win_used[b] = True
            return
        if not carried:
            self._site_done[b] = self._plain_done[b]
            self._plain_done[b].record(main)
        main.query()                 # submit the site pass before the host starts waiting
        self._launch_pending()       # the previous step's stage runs under this site pass
        self._pending = (b, index)

    # -- results ---------------------------------------------------------------------------

    def _sync(self) -> None:
        import torch

        self.flush()
        torch.cuda.current_stream(self.eng.device).synchronize()
        if self.side is not None:
            self.side.synchronize()

    def _wait_last_stage(self) -> None:
        """Wait until the last step's records and lists are on the host.  The plain form waits for the event
        behind ITS OWN windows stage (pass, stage and copies are ordered before it), not for the streams: several
        scorers whose passes queue on one stream -- the parts of a region, FeaturePreprocessor.score_and_write --
        hand their results over one after the other while the later passes still run."""
        b = (self._k - 1) % len(self._flags)
        if not self.overlap and self._k > 0 and self._win_used[b]:
            self._win_done[b].synchronize()
            return
        self._sync()

    def list_totals(self) -> list[tuple[int, int]]:
        """(entries of all U lists, entries of all Q lists) per set chunk, of the last step."""
        self._wait_last_stage()
        return [tuple(int(v) for v in ch.host_totals.tolist()) for ch in self.chunks]

    def results(self, grow: bool = False) -> WindowResults:
        """Synchronise and return the last step's records and candidate lists.  When the candidate
        buffers were too small: raise, or with ``grow=True`` enlarge them and repeat the windows
        stage on the per-site arrays of the last step (the genotypes are not streamed again)."""
        totals = self.list_totals()
        need_u, need_q = max(t[0] for t in totals), max(t[1] for t in totals)
        if need_u > self.cap_u or need_q > self.cap_q:
            if not grow:
                raise RuntimeError(
                    f"candidate buffers too small (need {need_u}/{need_q}); rebuild the scorer with larger cap_u/cap_q"
                )
            self._alloc_chunks(max(need_u, self.cap_u), max(need_q, self.cap_q))
            b = (self._k - 1) % len(self._flags)
            with self.window_stream():
                self._stage_plans[b].run()
                self._win_done[b].record(self.side)
            totals = self.list_totals()
        recs, offs, us, qs = [], [], [], []
        base_u = base_q = 0
        for ch, (nu, nq) in zip(self.chunks, totals):
            n_s = ch.s1 - ch.s0
            recs.append(np.frombuffer(ch.host_records.numpy().tobytes(), dtype=RECORD_DTYPE).reshape(n_s, self.n_windows))
            off = ch.host_offsets.numpy().reshape(n_s, self.n_windows, 2).copy()
            off[:, :, 0] += base_u
            off[:, :, 1] += base_q
            offs.append(off)
            for out, n, k in ((us, nu, 0), (qs, nq, 1)):
                if ch.host_lists is not None and n <= ch.n_fetch[k]:
                    out.append(ch.host_lists[k][:n].numpy().copy())  # came with the records
                else:
                    with self.window_stream():
                        out.append(ch.bufs[2 + k][:n].cpu().numpy())
            base_u += nu
            base_q += nq
        if len(self.chunks) == 1:
            return WindowResults(recs[0], offs[0], us[0], qs[0])
        return WindowResults(np.concatenate(recs), np.concatenate(offs), np.concatenate(us), np.concatenate(qs))

    # -- the row a rank contributes to the multi-GPU gather -------------------------------------

    def row_layout(self) -> "RowLayout":
        """Byte layout of this scorer's gather row, from the list sizes of the last step (they are
        fixed for a resident block): records of every set chunk, then all U lists, then all Q lists."""
        totals = self.list_totals()
        for nu, nq in totals:
            if nu > self.cap_u or nq > self.cap_q:
                raise RuntimeError(f"candidate buffers too small (need {nu}/{nq})")
        return RowLayout(self.n_sets, self.n_windows, [ch.s1 - ch.s0 for ch in self.chunks], [t[0] for t in totals],
                         [t[1] for t in totals])  # fmt: skip

    def pack_row(self, row, layout: "RowLayout") -> None:
        """Copy records and both candidate lists into the uint8 device tensor ``row`` (enqueued on
        the current stream -- call it on the stream the windows stage ran on)."""
        import torch

        o = 0
        for ch in self.chunks:
            row[o : o + ch.rec_bytes].copy_(ch.bufs[0], non_blocking=True)
            o += ch.rec_bytes
        for k, which in ((2, layout.n_u), (3, layout.n_q)):
            for ch, n in zip(self.chunks, which):
                row[o : o + 4 * n].copy_(ch.bufs[k][:n].view(torch.uint8), non_blocking=True)
                o += 4 * n
        assert o == layout.nbytes


@dataclass
class RowLayout:
    """What one rank sends to rank 0 per pass: ``n_sets * n_windows`` 24-byte records ([set][window],
    in chunks of at most SAI_MAX_SETS sets), then the int32 entries of all U candidate lists and of
    all Q candidate lists in (set, window) order -- fixed records + CSR lists; the list
    offsets are the prefix sums of the records' u_count / n_cdd_q, so they do not travel."""

    n_sets: int
    n_windows: int
    chunk_sets: list
    n_u: list
    n_q: list

    @property
    def nbytes(self) -> int:
        return self.n_sets * self.n_windows * RECORD_DTYPE.itemsize + 4 * (sum(self.n_u) + sum(self.n_q))

    def header(self) -> list[int]:
        return [self.n_sets, self.n_windows, len(self.chunk_sets), *self.chunk_sets, *self.n_u, *self.n_q]

    @classmethod
    def from_header(cls, h: Sequence[int]) -> "RowLayout":
        n_sets, n_windows, n_chunks = int(h[0]), int(h[1]), int(h[2])
        vals = [int(v) for v in h[3 : 3 + 3 * n_chunks]]
        return cls(n_sets, n_windows, vals[:n_chunks], vals[n_chunks : 2 * n_chunks], vals[2 * n_chunks :])

    def unpack(self, row: np.ndarray) -> WindowResults:
        """Rebuild the WindowResults of the sending rank from its row bytes (host side, rank 0)."""
        row = np.ascontiguousarray(row, dtype=np.uint8)
        if row.size != self.nbytes:
            raise ValueError(f"row of {row.size} bytes, layout says {self.nbytes}")
        rec_bytes = self.n_sets * self.n_windows * RECORD_DTYPE.itemsize
        rec = np.frombuffer(row[:rec_bytes].tobytes(), dtype=RECORD_DTYPE).reshape(self.n_sets, self.n_windows)
        tail = np.frombuffer(row[rec_bytes:].tobytes(), dtype=np.int32)
        cdd_u, cdd_q = tail[: sum(self.n_u)], tail[sum(self.n_u) :]
        off = np.zeros((self.n_sets, self.n_windows, 2), dtype=np.int64)
        for k, name in enumerate(("u_count", "n_cdd_q")):
            flat = rec[name].reshape(-1).astype(np.int64)
            off[:, :, k] = (np.cumsum(flat) - flat).reshape(self.n_sets, self.n_windows)
        if int(rec["u_count"].sum()) != cdd_u.size or int(rec["n_cdd_q"].sum()) != cdd_q.size:
            raise ValueError("row lists do not match the records' counts")
        return WindowResults(rec, off, cdd_u, cdd_q)


def default_windows(pos_first: int, pos_last: int, win_len: int, win_step: int) -> list[tuple]:
    """The reference's window grid over a chromosome (ChunkGenerator, chunk_generator.py:78)."""
    return split_genome([pos_first, pos_last], win_len, win_step)
