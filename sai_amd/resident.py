"""A chromosome block resident in HBM and its repeated scoring (the path bench.py times and the
multi-GPU driver shards): all buffers are allocated once, one ``step()`` enqueues the whole hot
path -- site_counts (+ fused per-site decision) -> window_bounds -> window_stats -> async copy of
the records to pinned host memory -- on the current HIP stream without any host synchronisation.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _ffi
from .engine import RECORD_DTYPE, Engine, TiledPop, WindowResults
from .utils.windows import split_genome


@dataclass
class ResidentBlock:
    """ref, tgt and source populations of one contiguous site range, plus positions."""

    pops: list  # [ref, tgt, src...] TiledPop
    ploidies: list  # same order
    pos: "object"  # int32 device tensor [n_sites]

    @property
    def n_sites(self) -> int:
        return self.pops[0].n_sites

    @property
    def genotype_bytes(self) -> int:
        """Algorithmic bytes of one site_counts launch: every genotype byte once."""
        return self.n_sites * sum(p.n_ind for p in self.pops)

    @property
    def packed2_bytes(self) -> int:
        """Algorithmic bytes of one packed2 launch: two bits per genotype, each site's row rounded
        up to whole bytes (the layout's padding of every site to 64-individual groups is not counted)."""
        return self.n_sites * sum((p.n_ind + 3) // 4 for p in self.pops)


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


class ResidentScorer:
    def __init__(self, eng: Engine, block: ResidentBlock, windows: Sequence[tuple], sets: Sequence[_ffi.SaiParams],
                 cap_u: int = 1 << 20, cap_q: int = 1 << 20, layout: str = "int8", overlap: bool = False):  # fmt: skip
        """``layout="packed2"`` re-encodes the block once into the 2-bit layout (dosages 0..2 only)
        and streams that instead: 4x fewer genotype bytes per step, identical results.

        ``overlap=True`` software-pipelines consecutive steps: the windows stage of step k
        (bounds, statistics, candidate lists, copy to the host) runs on a second HIP stream while
        the site pass of step k+1 already streams genotypes on the caller's stream; the per-site
        arrays are double-buffered and events order every reuse.  Same kernels, same results."""
        import torch

        if layout not in ("int8", "packed2"):
            raise ValueError("layout must be 'int8' or 'packed2'")
        self.layout = layout
        self.packed = [eng.pack2(p) for p in block.pops] if layout == "packed2" else None

        if not 1 <= len(sets) <= _ffi.SAI_MAX_SETS:
            raise ValueError(f"1..{_ffi.SAI_MAX_SETS} parameter sets per scorer")
        self.eng, self.block, self.sets = eng, block, list(sets)
        self.windows = list(windows)
        n, n_w, n_s = block.n_sites, len(self.windows), len(self.sets)
        self.n_windows, self.n_sets = n_w, n_s
        dev = eng.device
        self.win_start = torch.as_tensor(np.array([w[0] for w in self.windows], dtype=np.int64)).to(dev)
        self.win_end = torch.as_tensor(np.array([w[1] for w in self.windows], dtype=np.int64)).to(dev)
        # at most SAI_FUSED_SETS sets: one fused launch, the per-population counts never leave the chip
        self.fused = n_s <= _ffi.SAI_FUSED_SETS
        self.counts = None if self.fused else torch.empty((len(block.pops), n, 2), dtype=torch.int32, device=dev)
        self.overlap = bool(overlap)
        # two sets: the windows stage of step k runs under site pass k+1 and is over long before the
        # host may enqueue site pass k+2 into the same buffers
        n_buf = 2 if self.overlap else 1
        # the fused pass writes tgt_freq only at candidate sites (all the windows stage reads)
        self._tgt_freq = [torch.full((n,), float("nan"), dtype=torch.float64, device=dev) for _ in range(n_buf)]
        self._flags = [torch.empty((n_s, n), dtype=torch.uint8, device=dev) for _ in range(n_buf)]
        self.side = torch.cuda.Stream(device=dev, priority=-1) if self.overlap else None  # small kernels first
        self._site_done = [torch.cuda.Event() for _ in range(n_buf)]
        self._win_done = [torch.cuda.Event() for _ in range(n_buf)]  # after the windows stage that last read buffer b
        self._win_used = [False] * n_buf
        self._pending = None  # (buffer set, step index) whose windows stage has not been enqueued yet
        self.after_stage = None  # optional callable(step_index), run on the window stream right after a stage
        self._k = 0
        self.lo = torch.empty((n_w,), dtype=torch.int32, device=dev)
        self.hi = torch.empty((n_w,), dtype=torch.int32, device=dev)
        self.bufs = eng.alloc_window_bufs(n_s, n_w, cap_u, cap_q)
        # pinned mirror of the records | offsets | totals buffer: one copy per step
        self.host_head = torch.empty((self.bufs[5].numel(),), dtype=torch.uint8).pin_memory()
        rec_bytes = n_s * n_w * RECORD_DTYPE.itemsize
        self.host_records = self.host_head[:rec_bytes]
        self.host_offsets = self.host_head[rec_bytes:-16].view(torch.int64)
        self.host_totals = self.host_head[-16:].view(torch.int64)
        self.count_events: list = []  # (start, end) torch events around site_counts, when requested

    @property
    def tgt_freq(self):
        """Per-site buffers of the most recent step."""
        return self._tgt_freq[(self._k - 1) % len(self._tgt_freq)]

    @property
    def flags(self):
        return self._flags[(self._k - 1) % len(self._flags)]

    def window_stream(self):
        """Context manager selecting the stream on which window records are produced (for follow-up
        work such as the multi-GPU gather of ``bufs[0]``).  In the pipelined form call ``flush()``
        first, or use ``after_stage``: the newest step's stage is enqueued one step late."""
        import contextlib

        import torch

        return torch.cuda.stream(self.side) if self.overlap else contextlib.nullcontext()

    @staticmethod
    def _wait(event) -> None:
        while not event.query():  # polling: Event.synchronize() was seen to oversleep by ~7 ms
            pass

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
        self._wait(self._site_done[b])
        with torch.cuda.stream(self.side):
            self._window_stage(self._tgt_freq[b], self._flags[b])
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

        eng, blk = self.eng, self.block
        b = self._k % len(self._flags)
        tgt_freq, flags = self._tgt_freq[b], self._flags[b]
        main = torch.cuda.current_stream(eng.device)
        if self.overlap and self._win_used[b]:
            self._wait(self._win_done[b])  # the windows stage that last read buffer set b (2 steps ago) is done
        if time_counts:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.packed is not None:
            if self.fused:
                eng.site_pass_packed2(self.packed, blk.ploidies, self.sets, out=(tgt_freq, flags), freq_mode="candidates")
            else:
                eng.site_pass_packed2(self.packed, blk.ploidies, [], counts=self.counts)
        elif self.fused:
            eng.site_pass(blk.pops, blk.ploidies, self.sets, out=(tgt_freq, flags), freq_mode="candidates")
        else:
            eng.site_counts(blk.pops, out=self.counts)
        if time_counts:
            e1.record()
            self.count_events.append((e0, e1))
        if not self.fused:
            eng.site_flags(self.counts, blk.ploidies, self.sets, out=(tgt_freq, flags))
        index = self._k
        self._k += 1
        if not self.overlap:
            self._window_stage(tgt_freq, flags)
            if self.after_stage is not None:
                self.after_stage(index)
            return
        self._site_done[b].record(main)
        main.query()                 # submit the site pass before the host starts waiting
        self._launch_pending()       # the previous step's stage runs under this site pass
        self._pending = (b, index)

    def _window_stage(self, tgt_freq, flags) -> None:
        eng, blk = self.eng, self.block
        _ffi.check(
            eng.lib.sai_window_bounds(
                eng.ctx, eng._ptr(blk.pos), blk.n_sites, self.n_windows, eng._ptr(self.win_start), eng._ptr(self.win_end),
                eng._ptr(self.lo), eng._ptr(self.hi), eng._stream(),
            )
        )  # fmt: skip
        eng.window_stats_async(tgt_freq, flags, self.sets, self.lo, self.hi, blk.pos, self.bufs)
        self.host_head.copy_(self.bufs[5], non_blocking=True)

    def results(self) -> WindowResults:
        """Synchronise and return the last step's records and candidate lists."""
        import torch

        self.flush()
        torch.cuda.current_stream(self.eng.device).synchronize()
        if self.side is not None:
            self.side.synchronize()
        need_u, need_q = (int(v) for v in self.host_totals.tolist())
        if need_u > self.bufs[2].numel() or need_q > self.bufs[3].numel():
            raise RuntimeError(
                f"candidate buffers too small (need {need_u}/{need_q}); rebuild the scorer with larger cap_u/cap_q"
            )
        rec = np.frombuffer(self.host_records.numpy().tobytes(), dtype=RECORD_DTYPE).reshape(self.n_sets, self.n_windows)
        off = self.host_offsets.numpy().reshape(self.n_sets, self.n_windows, 2).copy()
        with self.window_stream():
            cdd_u, cdd_q = self.bufs[2][:need_u].cpu().numpy(), self.bufs[3][:need_q].cpu().numpy()
        return WindowResults(rec, off, cdd_u, cdd_q)


def default_windows(pos_first: int, pos_last: int, win_len: int, win_step: int) -> list[tuple]:
    """The reference's window grid over a chromosome (ChunkGenerator, chunk_generator.py:78)."""
    return split_genome([pos_first, pos_last], win_len, win_step)
