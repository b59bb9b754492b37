"""Host-side driver of libsaihip: device buffers (torch), streams, and the launch order.

torch is plumbing here (HBM allocations, the current HIP stream, H2D/D2H copies); every
statistic is computed by the HIP kernels behind the C ABI (include/saihip.h).  Nothing in this
module has a CPU path: without the built library or without a gfx950 device it raises.
"""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _ffi

RECORD_DTYPE = np.dtype(
    [("n_sites", "<i4"), ("u_count", "<i4"), ("n_cond", "<i4"), ("n_cdd_q", "<i4"), ("q", "<f8")]
)
assert RECORD_DTYPE.itemsize == C.sizeof(_ffi.SaiWindowRecord) == 24

FLAG_COND, FLAG_UCAND, FLAG_INVERTED = 1, 2, 4  # bits of flag_bytes()' bytes: condition, condition and tgt > x, site inverted
PLANES = _ffi.SAI_PLANES_PER_SET


def _torch():
    import torch

    return torch


def to_int8_dosage(gts) -> np.ndarray:
    """Narrow a reference-style genotype matrix ([sites][individuals], any integer dtype,
    negative = missing; utils.py:410) to C-contiguous int8.  Values inside the int8 range are kept
    as they are (DD works on the raw numbers); a value below -128 can only be a missing call for
    calc_freq (stat_utils.py:45-49) and is stored as -128; dosages above 127 cannot be represented
    and are rejected."""
    g = np.asarray(gts)
    if g.ndim != 2:
        raise ValueError("genotype matrix must be 2-D [sites][individuals]")
    if g.dtype == np.int8:
        return np.ascontiguousarray(g)
    if g.dtype == np.bool_:
        return np.ascontiguousarray(g.astype(np.int8))
    if not np.issubdtype(g.dtype, np.integer):
        raise TypeError(f"genotype matrix must have an integer dtype, got {g.dtype}")
    if g.size and g.strides[1] == g.itemsize and g.strides[0] > 0 and g.dtype.isnative:
        # one multithreaded native pass (the reference holds int64: numpy's max / maximum / astype
        # chain costs three passes over 8x the bytes)
        out = np.empty(g.shape, dtype=np.int8)
        lib = _ffi.load_host()
        rc = lib.sai_narrow_to_int8(
            g.ctypes.data_as(C.c_void_p), g.itemsize, int(np.issubdtype(g.dtype, np.signedinteger)), g.shape[0], g.shape[1],
            g.strides[0], out.ctypes.data_as(C.c_void_p), min(os.cpu_count() or 1, 16),
        )
        if rc == _ffi.SAI_ERR_UNSUPPORTED:
            raise ValueError("dosage above 127 is not representable in the int8 device layout")
        _ffi.check(rc, lib)
        return out
    if g.size and g.max() > 127:
        raise ValueError("dosage above 127 is not representable in the int8 device layout")
    if np.issubdtype(g.dtype, np.unsignedinteger):
        return np.ascontiguousarray(g.astype(np.int8))
    return np.ascontiguousarray(np.maximum(g, -128).astype(np.int8))


@dataclass
class TiledPop:
    """One population block resident in HBM in the tiled SoA layout of saihip.h."""

    tiles: "object"  # torch int8 tensor, 1-D
    n_sites: int
    n_ind: int


@dataclass
class PackedPop:
    """One population block in the optional packed2 layout of saihip.h (2 bits per call)."""

    data: "object"  # torch uint8 tensor, 1-D
    n_sites: int
    n_ind: int


@dataclass
class WindowResults:
    records: np.ndarray  # structured [n_sets][n_windows], RECORD_DTYPE
    offsets: np.ndarray  # int64 [n_sets][n_windows][2]
    cdd_u: np.ndarray  # int32 flat
    cdd_q: np.ndarray  # int32 flat
    # rows that answer several statistics from one parameter set point into the same lists (their offsets
    # are then not the running sums of their own counts); ``separate_lists()`` lays every row's lists out
    shared_lists: bool = False

    def separate_lists(self) -> "WindowResults":
        """The plain form -- the lists of every row behind each other in row order, offsets = running sums of
        the rows' counts -- which is what travels between ranks (WindowBatch.to_bytes)."""
        if not self.shared_lists:
            return self
        n_rows, n_w = self.records.shape
        parts = {0: [], 1: []}
        for i in range(n_rows):
            for col, flat, field in ((0, self.cdd_u, "u_count"), (1, self.cdd_q, "n_cdd_q")):
                a = int(self.offsets[i, 0, col]) if n_w else 0
                parts[col].append(flat[a : a + int(self.records[i][field].sum())])
        off = np.zeros((n_rows, n_w, 2), dtype=np.int64)
        for col, field in ((0, "u_count"), (1, "n_cdd_q")):
            counts = self.records[field].reshape(-1).astype(np.int64)
            off[:, :, col] = (np.cumsum(counts) - counts).reshape(n_rows, n_w)
        cat = lambda p, like: np.concatenate(p) if p else like[:0]  # noqa: E731
        return WindowResults(self.records, off, cat(parts[0], self.cdd_u), cat(parts[1], self.cdd_q))

    def u_list(self, s: int, w: int) -> np.ndarray:
        o = int(self.offsets[s, w, 0])
        return self.cdd_u[o : o + int(self.records[s, w]["u_count"])]

    def q_list(self, s: int, w: int) -> np.ndarray:
        o = int(self.offsets[s, w, 1])
        return self.cdd_q[o : o + int(self.records[s, w]["n_cdd_q"])]


class LaunchEvent:
    """A HIP event that a site-pass launch carries in its own dispatch packet (saihip.h,
    sai_plan_set_pass_events): no marker packet between two consecutive passes of a queue.  The part of
    torch.cuda.Event's interface the scorer and the bench use."""

    __slots__ = ("_lib", "_h")

    def __init__(self, eng: "Engine"):
        self._lib = eng.lib
        h = C.c_void_p()
        _ffi.check(eng.lib.sai_event_create(eng.ctx, C.byref(h)))
        self._h = h

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def synchronize(self) -> None:
        _ffi.check(self._lib.sai_event_synchronize(self._h))

    def query(self) -> bool:
        done = C.c_int32()
        _ffi.check(self._lib.sai_event_query(self._h, C.byref(done)))
        return bool(done.value)

    def elapsed_time(self, end: "LaunchEvent") -> float:
        ms = C.c_float()
        _ffi.check(self._lib.sai_event_elapsed_ms(self._h, end._h, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.sai_event_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class Plan:
    """A prepared launch sequence (saihip.h, sai_plan_*): the engine calls of a step recorded once
    with all their arguments, replayed by ONE C call on the current stream.  The tensors whose
    pointers the plan holds are kept alive here."""

    def __init__(self, eng: "Engine"):
        self.eng = eng
        self._keep: list = []
        h = C.c_void_p()
        _ffi.check(eng.lib.sai_plan_create(eng.ctx, C.byref(h)))
        self._h = h
        self._run = eng.lib.sai_plan_run

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.eng.lib.sai_plan_destroy(self._h)
            self._h = None
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def run(self) -> None:
        rc = self._run(self._h, self.eng._stream())
        if rc:
            _ffi.check(rc)

    def set_pass_events(self, start: "Optional[LaunchEvent]", stop: "Optional[LaunchEvent]") -> None:
        """The events every later ``run`` stamps in the dispatch packet of this plan's site pass."""
        self._events = (start, stop)  # kept alive
        _ffi.check(self.eng.lib.sai_plan_set_pass_events(self._h, start.handle if start is not None else None,
                                                         stop.handle if stop is not None else None))  # fmt: skip

    def _pops(self, pops, ploidies, packed: bool):
        arr = (_ffi.SaiPop * len(pops))()
        for i, p in enumerate(pops):
            t = p.data if packed else p.tiles
            arr[i].tiles = t.data_ptr() if t.numel() else 0
            arr[i].n_ind = p.n_ind
            arr[i].ploidy = int(ploidies[i]) if ploidies is not None else 1
            self._keep.append(t)
        return arr

    def add_site_counts(self, pops, counts) -> None:
        e = self.eng
        self._keep.append(counts)
        per_call = 2 + _ffi.SAI_FUSED_SRC  # populations a streaming pass takes: more go in groups
        for g0 in range(0, len(pops), per_call):
            part = pops[g0 : g0 + per_call]
            _ffi.check(e.lib.sai_plan_add_site_counts(self._h, part[0].n_sites, len(part), self._pops(part, None, False),
                                                      e._ptr(counts[g0 : g0 + len(part)])))  # fmt: skip

    def add_site_pass(self, pops, ploidies, sets, out, counts=None, freq_mode="dense", packed2=False, dd=None) -> None:
        """``Engine.site_pass`` / ``site_pass_packed2`` (``packed2=True``; ``sets == []`` = counts only);
        ``dd = (first source population, number of them, int32 tensor [2][rows][n_sites])``: DD's per-site
        terms ride along (``Engine.site_pass_dd``)."""
        e = self.eng
        n_sites = pops[0].n_sites
        pl_ptr, pl_stride = e._planes_arg(out[1], len(sets), n_sites) if sets else (None, PLANES * len(sets))
        self._keep.extend([t for t in (out or ()) if t is not None] + ([counts] if counts is not None else []))
        if dd is not None:
            if packed2:
                raise ValueError("DD rides along the int8 pass only")
            rows = e._dd_rows(pops, dd)
            self._keep.append(dd[2])
            _ffi.check(
                e.lib.sai_plan_add_site_pass_dd(
                    self._h, n_sites, len(pops), self._pops(pops, ploidies, False), e._ptr(counts) if counts is not None else None,
                    len(sets), e._params_array(sets) if sets else None, _ffi.FREQ_MODES[freq_mode],
                    e._ptr(out[0]) if sets else None, pl_ptr, pl_stride, C.byref(rows),
                )
            )  # fmt: skip
            return
        _ffi.check(
            e.lib.sai_plan_add_site_pass(
                self._h, n_sites, len(pops), self._pops(pops, ploidies, packed2), e._ptr(counts) if counts is not None else None,
                len(sets), e._params_array(sets) if sets else None, _ffi.FREQ_MODES[freq_mode],
                e._ptr(out[0]) if sets else None, pl_ptr, pl_stride, int(bool(packed2)),
            )
        )  # fmt: skip

    def add_site_flags(self, counts, ploidies, sets, out) -> None:
        e = self.eng
        n_pops, n_sites = int(counts.shape[0]), int(counts.shape[1])
        pl = (C.c_int32 * n_pops)(*[int(p) for p in ploidies])
        self._keep.extend([counts, out[0], out[1]])
        for s0 in range(0, len(sets), _ffi.SAI_MAX_SETS):
            chunk = sets[s0 : s0 + _ffi.SAI_MAX_SETS]
            pl_ptr, pl_stride = e._planes_arg(out[1][:, PLANES * s0 : PLANES * (s0 + len(chunk))], len(chunk), n_sites)
            _ffi.check(
                e.lib.sai_plan_add_site_flags(self._h, n_sites, n_pops, pl, e._ptr(counts), len(chunk), e._params_array(chunk),
                                              e._ptr(out[0]), pl_ptr, pl_stride)
            )  # fmt: skip

    def add_window_bounds(self, pos, win_start, win_end, seg_lo, seg_hi, lo, hi) -> None:
        e = self.eng
        self._keep.extend([t for t in (pos, win_start, win_end, seg_lo, seg_hi, lo, hi) if t is not None])
        _ffi.check(
            e.lib.sai_plan_add_window_bounds(
                self._h, e._ptr(pos), int(pos.numel()), int(lo.numel()), e._ptr(win_start), e._ptr(win_end),
                e._ptr(seg_lo) if seg_lo is not None else None, e._ptr(seg_hi) if seg_hi is not None else None, e._ptr(lo),
                e._ptr(hi),
            )
        )  # fmt: skip

    def add_window_stats(self, tgt_freq, planes, sets, lo, hi, pos, bufs) -> None:
        """``Engine.window_stats_async``."""
        e = self.eng
        n_sets, n_sites = len(sets), int(tgt_freq.numel())
        records, offsets, cdd_u, cdd_q, totals = bufs[:5]
        pl_ptr, pl_stride = e._planes_arg(planes, n_sets, n_sites)
        self._keep.extend([tgt_freq, planes, lo, hi, *bufs] + ([pos] if pos is not None else []))
        _ffi.check(
            e.lib.sai_plan_add_window_stats(
                self._h, n_sites, e._ptr(tgt_freq), pl_ptr, pl_stride, n_sets, e._params_array(sets), int(lo.numel()),
                e._ptr(lo), e._ptr(hi), e._ptr(pos) if pos is not None else None, e._ptr(records), e._ptr(offsets),
                e._ptr(cdd_u), int(cdd_u.numel()), e._ptr(cdd_q), int(cdd_q.numel()), e._ptr(totals),
            )
        )  # fmt: skip

    def add_copy_to_host(self, dst_pinned, src) -> None:
        if not dst_pinned.is_pinned():
            raise ValueError("the host side of a planned copy must be pinned memory")
        n = int(src.numel() * src.element_size())
        if int(dst_pinned.numel() * dst_pinned.element_size()) < n:
            raise ValueError("host buffer smaller than the device buffer")
        self._keep.extend([dst_pinned, src])
        _ffi.check(self.eng.lib.sai_plan_add_copy_to_host(self._h, C.c_void_p(dst_pinned.data_ptr()), self.eng._ptr(src), n))


class Engine:
    """One libsaihip context on one GPU of this process."""

    _cache: dict[int, "Engine"] = {}

    @classmethod
    def get(cls, device: Optional[int] = None) -> "Engine":
        torch = _torch()
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        if device not in cls._cache:
            cls._cache[device] = cls(device)
        return cls._cache[device]

    def __init__(self, device: int = 0):
        self.lib = _ffi.load()
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible to torch: sai_amd computes on MI355X only (no CPU fallback)")
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        ctx = C.c_void_p()
        _ffi.check(self.lib.sai_ctx_create(self.device_index, C.byref(ctx)))
        self.ctx = ctx
        self._tile_cache = None  # {id(matrix): (matrix, TiledPop)} while an upload_scope is open
        self._window_scope = None  # {"hints": ..., "results": ...} of the open upload_scope (stats/_window.py)
        self._scope_depth = 0
        self._staging = None  # pinned int8 buffer the host matrices are narrowed into
        self._staging_busy = None  # event behind the last H2D copies out of it
        self._pinned_free: dict[int, list] = {}  # capacity -> [(pinned uint8 tensor, [events])]

    def close(self) -> None:
        if getattr(self, "ctx", None):
            self.lib.sai_ctx_destroy(self.ctx)
            self.ctx = None
            self._pinned_free.clear()
            Engine._cache.pop(self.device_index, None)

    # -- helpers ---------------------------------------------------------------------------

    def identity(self) -> dict:
        """Which physical GPU this engine computes on: device index, PCI bus id, UUID, name -- what a
        multi-GPU record carries per rank (sai_device_identity)."""
        bus, uuid = C.create_string_buffer(32), C.create_string_buffer(40)
        _ffi.check(self.lib.sai_device_identity(self.device_index, bus, 32, uuid, 40))
        return {"device_index": self.device_index, "pci_bus_id": bus.value.decode(), "uuid": uuid.value.decode(),
                "name": _torch().cuda.get_device_name(self.device_index)}  # fmt: skip

    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    # Page-locking host memory is slow and erratic (measured on the MI355X box: 5 ms for a first
    # 400 KB buffer, 60-90 ms every third time torch's host allocator is asked again), so the pinned
    # mirrors of the window-stage buffers are recycled here instead of being pinned per scorer.
    _PINNED_KEEP = 8  # free buffers kept per capacity

    def pinned_acquire(self, nbytes: int):
        """A pinned uint8 host tensor of at least ``nbytes`` (capacity = next power of two >= 4 KiB).
        A recycled buffer is handed out only after the work recorded at its release has finished."""
        torch = _torch()
        cap = 1 << max(int(nbytes) - 1, 4095).bit_length()
        free = self._pinned_free.get(cap)
        if free:
            buf, events = free.pop()
            for ev in events:
                ev.synchronize()
            return buf
        return torch.empty((cap,), dtype=torch.uint8).pin_memory()

    def pinned_release(self, buf, streams=()) -> None:
        """Give a buffer of ``pinned_acquire`` back; copies still in flight on ``streams`` (and on the
        current stream) are waited for when the buffer is handed out again."""
        torch = _torch()
        free = self._pinned_free.setdefault(int(buf.numel()), [])
        if len(free) >= self._PINNED_KEEP:
            return
        events = []
        for st in (torch.cuda.current_stream(self.device), *streams):
            if st is not None:
                ev = torch.cuda.Event()
                ev.record(st)
                events.append(ev)
        free.append((buf, events))

    def _empty(self, shape, dtype):
        return _torch().empty(shape, dtype=dtype, device=self.device)

    @staticmethod
    def _ptr(t) -> C.c_void_p:
        return C.c_void_p(t.data_ptr() if t is not None and t.numel() else 0)

    def _params_array(self, sets: Sequence[_ffi.SaiParams]):
        arr = (_ffi.SaiParams * len(sets))()
        for i, s in enumerate(sets):
            arr[i] = s
        return arr

    def plan(self) -> Plan:
        return Plan(self)

    def release_ingest_buffers(self) -> None:
        """Drop the staging the VCF readers keep between calls (utils.device_vcf): two pinned compressed
        buffers, two text buffers in HBM and the line tables -- about 650 MB of HBM and 125 MB of
        page-locked host memory at the 248 MiB batch of a large bgzip file.  They are kept because
        allocating and page-locking them costs 16-90 ms per read; ``score`` releases them when it is
        done, a caller that reads region after region keeps them."""
        torch = _torch()
        for key in ("_inflate_state", "_ingest_state"):
            st = self.__dict__.pop(key, None)
            if st:
                for name in ("side", "copy", "d2h", "tok", "stream"):
                    if name in st:
                        st[name].synchronize()
                st.clear()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()

    # -- layout ----------------------------------------------------------------------------

    def upload_scope(self, hints=None):
        """Context manager: inside it, a host matrix handed to ``tile`` more than once (the same
        object: FeaturePreprocessor.run passes one window's matrices to every configured statistic)
        is narrowed, uploaded and re-tiled ONCE.  The caller promises not to modify the matrices
        while the scope is open; nothing is remembered after it closes.  ``hints`` = {(ploidies, w, y_list,
        anc): {"x": ..., "quantile": ...}}: which U and Q thresholds of the window belong to one parameter set,
        so that the two statistics share ONE device call (stats/_window.py)."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            if self._scope_depth == 0:
                self._tile_cache = {}
                self._window_scope = {"hints": dict(hints or {}), "results": {}}
            self._scope_depth += 1
            try:
                yield self
            finally:
                self._scope_depth -= 1
                if self._scope_depth == 0:
                    self._tile_cache = None
                    self._window_scope = None

        return scope()

    def window_scope(self):
        return self._window_scope

    def _stage_host(self, g: np.ndarray, offset: int):
        """Narrow a host matrix to int8 [sites][individuals] straight into the engine's pinned
        staging buffer at ``offset`` (one multithreaded native pass); returns the torch view, or None
        when the matrix needs the general numpy path (bools, column-strided views, ...)."""
        native = (g.size and np.issubdtype(g.dtype, np.integer) and g.dtype != np.bool_ and g.strides[1] == g.itemsize
                  and g.strides[0] > 0 and g.dtype.isnative)  # fmt: skip
        if not native:
            return None
        out = self._staging[offset : offset + g.size].view(g.shape[0], g.shape[1])
        lib = _ffi.load_host()
        rc = lib.sai_narrow_to_int8(
            g.ctypes.data_as(C.c_void_p), g.itemsize, int(np.issubdtype(g.dtype, np.signedinteger)), g.shape[0], g.shape[1],
            g.strides[0], C.c_void_p(out.data_ptr()), min(os.cpu_count() or 1, 16),
        )  # fmt: skip
        if rc == _ffi.SAI_ERR_UNSUPPORTED:
            raise ValueError("dosage above 127 is not representable in the int8 device layout")
        _ffi.check(rc, lib)
        return out

    def tile(self, gts) -> TiledPop:
        """Upload a [sites][individuals] matrix (host numpy, any int dtype, or a device int8
        tensor) and re-tile it on the GPU."""
        return self.tile_many([gts])[0]

    def tile_many(self, mats: Sequence) -> list:
        """``tile`` for several matrices with ONE host synchronisation: every host matrix is narrowed
        into its own part of the pinned staging buffer, the H2D copies and the re-tiling kernels are
        enqueued back to back, and the host waits once at the end (the staging buffer must not be
        rewritten before the copies have left it).  Matrices already seen in the open
        ``upload_scope`` are not uploaded again."""
        torch = _torch()
        cache = self._tile_cache
        out: list = [None] * len(mats)
        host = []  # (index, array)
        for i, m in enumerate(mats):
            if isinstance(m, torch.Tensor):
                if m.dtype != torch.int8 or m.dim() != 2:
                    raise TypeError("device genotype matrix must be a 2-D int8 tensor")
                out[i] = self._tile_device(m.to(self.device).contiguous())
            elif cache is not None and id(m) in cache:
                out[i] = cache[id(m)][1]
            else:
                g = np.asarray(m)
                if g.ndim != 2:
                    raise ValueError("genotype matrix must be 2-D [sites][individuals]")
                host.append((i, g))
        if not host:
            return out
        total = sum(int(g.size) for _, g in host)
        # the staging buffer must not be rewritten before the copies of the call before have left it: an event
        # behind them is waited for HERE -- by then it has long passed -- instead of a stream synchronisation at
        # the end of every call (one host wait less per window of the per-window route)
        if self._staging_busy is not None:
            self._staging_busy.synchronize()
            self._staging_busy = None
        if self._staging is None or self._staging.numel() < total:
            self._staging = torch.empty((max(int(total * 1.25), 1 << 20),), dtype=torch.int8).pin_memory()
        offset, staged = 0, False
        for i, g in host:
            view = self._stage_host(g, offset)
            if view is None:
                src = torch.from_numpy(to_int8_dosage(g)).to(self.device)
            else:
                src = view.to(self.device, non_blocking=True)
                offset += int(g.size)
                staged = True
            out[i] = self._tile_device(src)
            if cache is not None:
                cache[id(mats[i])] = (mats[i], out[i])  # holding the matrix keeps its id from being reused
        if staged:
            self._staging_busy = torch.cuda.Event()
            self._staging_busy.record(torch.cuda.current_stream(self.device))
        return out

    def tile_columns(self, block, cols: Sequence[int]) -> TiledPop:
        """Tiled block of the columns ``cols`` of a device int8 [records][samples] block (what the GPU
        tokenizer leaves behind): a contiguous run of columns is re-tiled in place with the block's
        row stride, anything else is gathered first."""
        torch = _torch()
        cols = [int(c) for c in cols]
        n_sites, width = int(block.shape[0]), int(block.shape[1])
        if cols and cols == list(range(cols[0], cols[0] + len(cols))):
            n_ind = len(cols)
            nbytes = self.lib.sai_tiled_bytes(n_sites, n_ind)
            dst = self._empty((max(nbytes, 0),), torch.int8)
            src_ptr = C.c_void_p(block.data_ptr() + cols[0]) if block.numel() else C.c_void_p(0)
            _ffi.check(self.lib.sai_tile_from_site_major(self.ctx, src_ptr, n_sites, n_ind, width, self._ptr(dst), self._stream()))
            return TiledPop(dst, n_sites, n_ind)
        idx = torch.tensor(cols, dtype=torch.int64, device=self.device)
        return self._tile_device(block.index_select(1, idx).contiguous())

    def _tile_device(self, src) -> TiledPop:
        torch = _torch()
        n_sites, n_ind = int(src.shape[0]), int(src.shape[1])
        nbytes = self.lib.sai_tiled_bytes(n_sites, n_ind)
        dst = self._empty((max(nbytes, 0),), torch.int8)
        _ffi.check(
            self.lib.sai_tile_from_site_major(
                self.ctx, self._ptr(src), n_sites, n_ind, n_ind, self._ptr(dst), self._stream()
            )
        )
        return TiledPop(dst, n_sites, n_ind)

    # -- kernels ---------------------------------------------------------------------------

    def single_window(self, pops: Sequence[TiledPop], ploidies: Sequence[int], prm: _ffi.SaiParams):
        """U and Q of ONE window = the whole blocks (sai_single_window: one C call, one
        synchronisation).  Returns (record, U site indices, Q site indices) on the host."""
        n_sites = pops[0].n_sites
        if any(p.n_sites != n_sites for p in pops):
            raise ValueError("all populations of one call must cover the same sites")
        arr = (_ffi.SaiPop * len(pops))()
        for i, p in enumerate(pops):
            arr[i].tiles = p.tiles.data_ptr() if p.tiles.numel() else 0
            arr[i].n_ind = p.n_ind
            arr[i].ploidy = int(ploidies[i])
        rec = _ffi.SaiWindowRecord()
        cdd_u = np.empty(max(n_sites, 1), dtype=np.int32)
        cdd_q = np.empty(max(n_sites, 1), dtype=np.int32)
        _ffi.check(
            self.lib.sai_single_window(self.ctx, n_sites, len(pops), arr, C.byref(prm), C.byref(rec),
                                       cdd_u.ctypes.data_as(C.c_void_p), cdd_q.ctypes.data_as(C.c_void_p), self._stream())
        )  # fmt: skip
        return rec, cdd_u[: rec.u_count].astype(np.int64), cdd_q[: rec.n_cdd_q].astype(np.int64)

    def site_counts(self, pops: Sequence[TiledPop], out=None):
        """{alt_sum, n_called} per population and site: int32 tensor [P][n_sites][2]."""
        torch = _torch()
        n_sites = pops[0].n_sites
        if any(p.n_sites != n_sites for p in pops):
            raise ValueError("all populations of one call must cover the same sites")
        if out is None:
            out = self._empty((len(pops), n_sites, 2), torch.int32)
        per_call = 2 + _ffi.SAI_FUSED_SRC  # populations a streaming pass takes: more go in groups
        for g0 in range(0, len(pops), per_call):
            part = pops[g0 : g0 + per_call]
            arr = (_ffi.SaiPop * len(part))()
            for i, p in enumerate(part):
                arr[i].tiles = p.tiles.data_ptr() if p.tiles.numel() else 0
                arr[i].n_ind = p.n_ind
                arr[i].ploidy = 1
            _ffi.check(self.lib.sai_site_counts(self.ctx, n_sites, len(part), arr, self._ptr(out[g0 : g0 + len(part)]), self._stream()))
        return out

    def alloc_planes(self, n_sites: int, n_sets: int):
        """Flag planes of ``n_sets`` parameter sets (saihip.h): int64 tensor [tiles][3 * n_sets], zeroed.
        A call with n sets uses words 0 ("any": sites whose tgt_freq is stored), 1 + s (condition of set s)
        and, when a set lacks ancestral alleles, 1 + n + s (site inverted) of a row; bit b = site 64 t + b.
        Sets beyond SAI_MAX_SETS are evaluated in several calls: the column slice
        ``planes[:, 3 * s0 : 3 * s1]`` is the row space of the call for sets s0..s1-1."""
        torch = _torch()
        n_tiles = (int(n_sites) + _ffi.SAI_TILE_SITES - 1) // _ffi.SAI_TILE_SITES
        return torch.zeros((n_tiles, PLANES * int(n_sets)), dtype=torch.int64, device=self.device)

    @staticmethod
    def _planes_arg(planes, n_sets: int, n_sites: int):
        """(pointer, row stride in words) of a planes tensor or column slice of one."""
        n_tiles = (int(n_sites) + _ffi.SAI_TILE_SITES - 1) // _ffi.SAI_TILE_SITES
        if planes.dim() != 2 or planes.element_size() != 8 or planes.shape[0] != n_tiles or planes.shape[1] != PLANES * n_sets:
            raise ValueError(f"flag planes must be int64 [{n_tiles}][{PLANES * n_sets}], got {tuple(planes.shape)}")
        if n_tiles > 1 and planes.shape[1] and (planes.stride(1) != 1 or planes.stride(0) < planes.shape[1]):
            raise ValueError("flag planes must be rows of consecutive words")
        stride = int(planes.stride(0)) if n_tiles > 1 else max(int(planes.stride(0)), PLANES * n_sets)
        return C.c_void_p(planes.data_ptr() if planes.numel() else 0), stride

    @staticmethod
    def _row_words(sets, s0: int, s1: int):
        """(word of "any", words of the conditions, words of the inverted planes or None) of the call that
        evaluated sets s0..s1-1 of ``sets``, as columns of the whole planes tensor."""
        base, n = PLANES * s0, s1 - s0
        with_inv = any(not p.anc_allele_available for p in sets[s0:s1])
        return base, [base + 1 + k for k in range(n)], ([base + 1 + n + k for k in range(n)] if with_inv else None)

    def flag_bytes(self, planes, n_sites: int, sets, tgt_freq=None):
        """The planes as one byte per set and site (uint8 tensor [n_sets][n_sites]; FLAG_COND |
        FLAG_INVERTED) -- for tests and for the single-window helpers, not on the hot path.  ``sets`` = the
        parameter sets the planes were written for (they decide whether inverted words exist).  With the
        pass's ``tgt_freq`` the bytes also carry FLAG_UCAND = condition and effective target frequency > x
        (u_statistic.py:92), evaluated here in torch f64 the way the windows stage evaluates it."""
        torch = _torch()
        n_tiles, n_sets = int(planes.shape[0]), len(sets)
        if int(planes.shape[1]) != PLANES * n_sets:
            raise ValueError(f"planes of {planes.shape[1] // PLANES} sets, {n_sets} parameter sets")
        shifts = torch.arange(_ffi.SAI_TILE_SITES, dtype=torch.int64, device=planes.device)
        out = torch.zeros((n_sets, n_tiles * _ffi.SAI_TILE_SITES), dtype=torch.uint8, device=planes.device)
        for s0 in range(0, n_sets, _ffi.SAI_MAX_SETS):
            s1 = min(s0 + _ffi.SAI_MAX_SETS, n_sets)
            _, cond, inv = self._row_words(sets, s0, s1)
            for k, s in enumerate(range(s0, s1)):
                out[s] |= (((planes[:, cond[k]].unsqueeze(1) >> shifts) & 1).reshape(-1) * FLAG_COND).to(torch.uint8)
                if inv is not None:
                    out[s] |= (((planes[:, inv[k]].unsqueeze(1) >> shifts) & 1).reshape(-1) * FLAG_INVERTED).to(torch.uint8)
        out = out[:, : int(n_sites)].contiguous()
        if tgt_freq is not None:
            f = self.site_tgt_freq(planes, tgt_freq, n_sites)
            for s in range(n_sets):
                eff = torch.where((out[s] & FLAG_INVERTED) != 0, 1.0 - f, f)
                out[s] |= (((out[s] & FLAG_COND) != 0) & (eff > float(sets[s].x))).to(torch.uint8) * FLAG_UCAND
        return out

    def site_tgt_freq(self, planes, tgt_freq, n_sites: int):
        """tgt_freq per SITE (f64 [n_sites], NaN where the pass stored nothing) from the packed slots of the
        first call's "any" word (saihip.h) -- for tests; the windows stage reads the slots directly."""
        torch = _torch()
        n_tiles = int(planes.shape[0])
        shifts = torch.arange(_ffi.SAI_TILE_SITES, dtype=torch.int64, device=planes.device)
        stored = ((planes[:, 0].unsqueeze(1) >> shifts) & 1).to(torch.int64)  # [tiles][64]
        slot = torch.cumsum(stored, dim=1) - stored + torch.arange(n_tiles, device=planes.device).unsqueeze(1) * _ffi.SAI_TILE_SITES
        padded = torch.full((n_tiles * _ffi.SAI_TILE_SITES,), float("nan"), dtype=torch.float64, device=planes.device)
        padded[: int(tgt_freq.numel())] = tgt_freq
        out = torch.where(stored.bool(), padded[slot.clamp(max=padded.numel() - 1)], torch.full_like(padded, float("nan")).reshape(n_tiles, -1))
        return out.reshape(-1)[: int(n_sites)].contiguous()

    def _pass_out(self, n_sites: int, n_sets: int, freq_mode: str):
        torch = _torch()
        if freq_mode == "candidates":  # entries the pass does not write read as NaN, never as garbage
            freq = torch.full((n_sites,), float("nan"), dtype=torch.float64, device=self.device)
        else:
            freq = self._empty((n_sites,), torch.float64)
        return freq, self.alloc_planes(n_sites, n_sets)

    def site_pass(self, pops: Sequence[TiledPop], ploidies: Sequence[int], sets: Sequence[_ffi.SaiParams], out=None,
                  counts=None, freq_mode: str = "dense"):
        """Fused site_counts + site_flags (at most SAI_FUSED_SETS parameter sets): one launch,
        the per-population counts stay on chip unless a ``counts`` tensor is passed.  Returns
        (tgt_freq, flag planes) exactly as ``site_flags(site_counts(pops), ...)`` would; with
        ``freq_mode="candidates"`` tgt_freq is written only where some set's condition bit is up
        (all that ``window_stats`` reads) and every other entry of ``out[0]`` is left as it was."""
        torch = _torch()
        n_sites = pops[0].n_sites
        if any(p.n_sites != n_sites for p in pops):
            raise ValueError("all populations of one call must cover the same sites")
        if len(sets) > _ffi.SAI_FUSED_SETS:
            raise ValueError(f"site_pass carries at most {_ffi.SAI_FUSED_SETS} parameter sets")
        arr = (_ffi.SaiPop * len(pops))()
        for i, p in enumerate(pops):
            arr[i].tiles = p.tiles.data_ptr() if p.tiles.numel() else 0
            arr[i].n_ind = p.n_ind
            arr[i].ploidy = int(ploidies[i])
        if out is None:
            out = self._pass_out(n_sites, len(sets), freq_mode)
        pl_ptr, pl_stride = self._planes_arg(out[1], len(sets), n_sites)
        _ffi.check(
            self.lib.sai_site_pass(
                self.ctx, n_sites, len(pops), arr, self._ptr(counts) if counts is not None else None, len(sets),
                self._params_array(sets), _ffi.FREQ_MODES[freq_mode], self._ptr(out[0]), pl_ptr, pl_stride,
                self._stream(),
            )
        )  # fmt: skip
        return out

    @staticmethod
    def dd_rides_along(pops: Sequence[TiledPop], first: int, n_src_pops: int) -> bool:
        """Whether ``site_pass_dd`` serves these populations: at most SAI_DD_FUSED_ROWS source individuals in all."""
        rows = sum(p.n_ind for p in pops[first : first + n_src_pops])
        return 1 <= rows <= _ffi.SAI_DD_FUSED_ROWS

    def _dd_rows(self, pops, dd) -> "_ffi.SaiDdRows":
        first, n_src_pops, out = dd
        rows = sum(p.n_ind for p in pops[first : first + n_src_pops])
        if tuple(out.shape) != (2, rows, pops[0].n_sites) or out.element_size() != 4 or not out.is_contiguous():
            raise ValueError(f"DD terms go to a contiguous int32 tensor [2][{rows}][{pops[0].n_sites}]")
        r = _ffi.SaiDdRows()
        r.first_pop, r.n_pops, r.absdiff = int(first), int(n_src_pops), out.data_ptr() if out.numel() else 0
        return r

    def site_pass_dd(self, pops: Sequence[TiledPop], ploidies: Sequence[int], sets: Sequence[_ffi.SaiParams], first_src: int,
                     n_src_pops: int, out=None, counts=None, freq_mode: str = "dense", absdiff=None):
        """``site_pass`` with DD's per-site terms riding along (sai_site_pass_dd): returns (out, absdiff) with
        absdiff = int32 [2][rows][n_sites] -- [0] against ``pops[0]``, [1] against ``pops[1]``, row = the
        individuals of ``pops[first_src : first_src + n_src_pops]`` in order.  ``sets == []``: counts (``counts``
        must be given) and DD only; the populations behind tgt are then merely counted."""
        torch = _torch()
        n_sites = pops[0].n_sites
        if any(p.n_sites != n_sites for p in pops):
            raise ValueError("all populations of one call must cover the same sites")
        arr = (_ffi.SaiPop * len(pops))()
        for i, p in enumerate(pops):
            arr[i].tiles = p.tiles.data_ptr() if p.tiles.numel() else 0
            arr[i].n_ind = p.n_ind
            arr[i].ploidy = int(ploidies[i]) if ploidies is not None else 1
        if out is None and sets:
            out = self._pass_out(n_sites, len(sets), freq_mode)
        rows = sum(p.n_ind for p in pops[first_src : first_src + n_src_pops])
        if absdiff is None:
            absdiff = self._empty((2, rows, n_sites), torch.int32)
        dd = self._dd_rows(pops, (first_src, n_src_pops, absdiff))
        pl_ptr, pl_stride = self._planes_arg(out[1], len(sets), n_sites) if sets else (None, PLANES * len(sets))
        _ffi.check(
            self.lib.sai_site_pass_dd(
                self.ctx, n_sites, len(pops), arr, self._ptr(counts) if counts is not None else None, len(sets),
                self._params_array(sets) if sets else None, _ffi.FREQ_MODES[freq_mode], self._ptr(out[0]) if sets else None,
                pl_ptr, pl_stride, C.byref(dd), self._stream(),
            )
        )  # fmt: skip
        return out, absdiff

    def pack2(self, pop: TiledPop) -> PackedPop:
        """Re-encode a tiled int8 block (dosages 0..2, negative = missing) as packed2.  Raises if a
        dosage above 2 is present (polyploid data stay on the int8 path)."""
        torch = _torch()
        nbytes = self.lib.sai_packed2_bytes(pop.n_sites, pop.n_ind)
        if nbytes < 0:
            raise ValueError("packed2: population too large")
        data = self._empty((nbytes,), torch.uint8)
        bad = self._empty((1,), torch.int32)
        _ffi.check(
            self.lib.sai_pack2_from_tiles(
                self.ctx, self._ptr(pop.tiles), pop.n_sites, pop.n_ind, self._ptr(data), self._ptr(bad), self._stream()
            )
        )
        if int(bad.item()) != 0:
            raise ValueError("dosage above 2: this block cannot be held in the packed2 layout")
        return PackedPop(data, pop.n_sites, pop.n_ind)

    def site_pass_packed2(self, pops: Sequence[PackedPop], ploidies: Sequence[int], sets: Sequence[_ffi.SaiParams],
                          out=None, counts=None, freq_mode: str = "dense"):
        """``site_pass`` on packed2 blocks; with ``sets == []`` only the counts are produced."""
        torch = _torch()
        n_sites = pops[0].n_sites
        if any(p.n_sites != n_sites for p in pops):
            raise ValueError("all populations of one call must cover the same sites")
        arr = (_ffi.SaiPop * len(pops))()
        for i, p in enumerate(pops):
            arr[i].tiles = p.data.data_ptr() if p.data.numel() else 0
            arr[i].n_ind = p.n_ind
            arr[i].ploidy = int(ploidies[i])
        if out is None and sets:
            out = self._pass_out(n_sites, len(sets), freq_mode)
        pl_ptr, pl_stride = self._planes_arg(out[1], len(sets), n_sites) if out else (None, PLANES * len(sets))
        _ffi.check(
            self.lib.sai_site_pass_packed2(
                self.ctx, n_sites, len(pops), arr, self._ptr(counts) if counts is not None else None, len(sets),
                self._params_array(sets) if sets else None, _ffi.FREQ_MODES[freq_mode],
                self._ptr(out[0]) if out else None, pl_ptr, pl_stride, self._stream(),
            )
        )  # fmt: skip
        return out

    def site_flags(self, counts, ploidies: Sequence[int], sets: Sequence[_ffi.SaiParams], want_adj=False, out=None):
        """(tgt_freq f64 [n], flag planes i64 [tiles][3 S] (``alloc_planes``), adj f64 [S][2][n] or None)."""
        torch = _torch()
        n_pops, n_sites = int(counts.shape[0]), int(counts.shape[1])
        pl = (C.c_int32 * n_pops)(*[int(p) for p in ploidies])
        n_sets = len(sets)
        if out is None:
            tgt_freq = self._empty((n_sites,), torch.float64)
            planes = self.alloc_planes(n_sites, n_sets)
        else:
            tgt_freq, planes = out
        adj = self._empty((n_sets, 2, n_sites), torch.float64) if want_adj else None
        for s0 in range(0, n_sets, _ffi.SAI_MAX_SETS):
            chunk = sets[s0 : s0 + _ffi.SAI_MAX_SETS]
            pl_ptr, pl_stride = self._planes_arg(planes[:, PLANES * s0 : PLANES * (s0 + len(chunk))], len(chunk), n_sites)
            _ffi.check(
                self.lib.sai_site_flags(
                    self.ctx, n_sites, n_pops, pl, self._ptr(counts), len(chunk), self._params_array(chunk),
                    self._ptr(tgt_freq), pl_ptr, pl_stride, self._ptr(adj[s0:]) if want_adj else None,
                    self._stream(),
                )
            )  # fmt: skip
        return tgt_freq, planes, adj

    def site_freqs(self, counts, ploidies: Sequence[int]):
        """f64 frequency per population and site ([P][n_sites], NaN where nothing is called)."""
        torch = _torch()
        n_pops, n_sites = int(counts.shape[0]), int(counts.shape[1])
        pl = (C.c_int32 * n_pops)(*[int(p) for p in ploidies])
        freqs = self._empty((n_pops, n_sites), torch.float64)
        _ffi.check(self.lib.sai_site_freqs(self.ctx, n_sites, n_pops, pl, self._ptr(counts), self._ptr(freqs), self._stream()))
        return freqs

    def window_fourpop(self, freqs, n_src: int, has_outgroup: bool, lo, hi):
        """fd, df, Danc, Dplus per (window, source): f64 tensor [n_windows][n_src][4]; ``freqs`` =
        [ref, tgt, sources..., (outgroup)] from ``site_freqs``."""
        torch = _torch()
        n_w = int(lo.numel())
        sums = self._empty((n_w, n_src, 7), torch.float64)
        stats = self._empty((n_w, n_src, 4), torch.float64)
        _ffi.check(
            self.lib.sai_window_fourpop(
                self.ctx, int(freqs.shape[1]), n_src, 1 if has_outgroup else 0, self._ptr(freqs), n_w, self._ptr(lo),
                self._ptr(hi), self._ptr(sums), self._ptr(stats), self._stream(),
            )
        )  # fmt: skip
        return stats

    def fourpop_windows(self, counts, ploidies: Sequence[int], n_src: int, has_outgroup: bool, lo, hi):
        """fd, df, Danc, Dplus per (window, source) for ANY number of sources: ``counts`` = int32 [P][n_sites][2] in
        the order ref, tgt, sources..., (outgroup).  Every source is a statistic of its own (fd_statistic.py:63-88),
        so the sources go through ``site_freqs`` / ``window_fourpop`` SAI_FUSED_SRC at a time."""
        torch = _torch()
        parts = []
        tail = [2 + n_src] if has_outgroup else []
        for s0 in range(0, n_src, _ffi.SAI_FUSED_SRC):
            s1 = min(s0 + _ffi.SAI_FUSED_SRC, n_src)
            rows = [0, 1, *range(2 + s0, 2 + s1), *tail]
            sel = counts if rows == list(range(int(counts.shape[0]))) else counts[torch.tensor(rows, device=counts.device)]
            freqs = self.site_freqs(sel, [ploidies[r] for r in rows])
            parts.append(self.window_fourpop(freqs, s1 - s0, has_outgroup, lo, hi))
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)

    def single_window_unfused(self, pops: Sequence[TiledPop], ploidies: Sequence[int], prm: _ffi.SaiParams):
        """``single_window`` for more than SAI_FUSED_SRC source populations: counts in groups, the stand-alone
        per-site decision (which takes up to SAI_MAX_SRC sources), the window statistics over [0, n_sites)."""
        torch = _torch()
        n_sites = pops[0].n_sites
        rec = _ffi.SaiWindowRecord()
        if n_sites == 0:
            rec.q = float("nan")
            return rec, np.zeros(0, np.int64), np.zeros(0, np.int64)
        counts = self.site_counts(pops)
        tgt_freq, planes, _ = self.site_flags(counts, ploidies, [prm])
        lo = torch.zeros(1, dtype=torch.int32, device=self.device)
        hi = torch.full((1,), n_sites, dtype=torch.int32, device=self.device)
        res = self.window_stats(tgt_freq, planes, [prm], lo, hi, pos=None, cap_hint=max(n_sites, 1))
        r = res.records[0, 0]
        rec.n_sites, rec.u_count, rec.n_cond, rec.n_cdd_q, rec.q = int(r["n_sites"]), int(r["u_count"]), int(r["n_cond"]), int(r["n_cdd_q"]), float(r["q"])
        return rec, res.u_list(0, 0).astype(np.int64), res.q_list(0, 0).astype(np.int64)

    def pattern_sum(self, ref_freq, tgt_freq, src_freq, out_freq, pattern_bits: int) -> float:
        """np.sum over the sites of the per-site pattern product (calc_pattern_sum's arithmetic) of
        four f64 frequency arrays (host numpy or device tensors of one length)."""
        torch = _torch()
        dev = []
        for f in (ref_freq, tgt_freq, src_freq, out_freq):
            t = f if isinstance(f, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(f, dtype=np.float64))
            dev.append(t.to(self.device, dtype=torch.float64).contiguous())
        n = int(dev[0].numel())
        if any(int(t.numel()) != n for t in dev):
            raise ValueError("frequency arrays must have the same length")
        out = self._empty((1,), torch.float64)
        _ffi.check(
            self.lib.sai_pattern_sum(self.ctx, n, self._ptr(dev[0]), self._ptr(dev[1]), self._ptr(dev[2]), self._ptr(dev[3]),
                                     int(pattern_bits), self._ptr(out), self._stream())
        )  # fmt: skip
        return float(out.item())

    def site_absdiff(self, pop: TiledPop, src: TiledPop):
        """int32 [src.n_ind][n_sites]: per site, sum over pop's individuals of |src - g| (DD's
        per-site city-block terms)."""
        torch = _torch()
        if pop.n_sites != src.n_sites:
            raise ValueError("populations must cover the same sites")
        a, b = _ffi.SaiPop(), _ffi.SaiPop()
        a.tiles, a.n_ind, a.ploidy = (pop.tiles.data_ptr() if pop.tiles.numel() else 0), pop.n_ind, 1
        b.tiles, b.n_ind, b.ploidy = (src.tiles.data_ptr() if src.tiles.numel() else 0), src.n_ind, 1
        out = self._empty((src.n_ind, pop.n_sites), torch.int32)
        _ffi.check(self.lib.sai_site_absdiff(self.ctx, pop.n_sites, C.byref(a), C.byref(b), self._ptr(out), self._stream()))
        return out

    def window_dd(self, ad_ref, n_ref_ind: int, ad_tgt, n_tgt_ind: int, lo, hi):
        """f64 [n_windows]: DD of one source population from its per-site terms."""
        torch = _torch()
        n_src_ind, n_sites = int(ad_ref.shape[0]), int(ad_ref.shape[1])
        n_w = int(lo.numel())
        scratch = self._empty((n_w, n_src_ind), torch.float64)
        dd = self._empty((n_w,), torch.float64)
        _ffi.check(
            self.lib.sai_window_dd(
                self.ctx, n_sites, n_src_ind, self._ptr(ad_ref), n_ref_ind, self._ptr(ad_tgt), n_tgt_ind, n_w,
                self._ptr(lo), self._ptr(hi), self._ptr(scratch), self._ptr(dd), self._stream(),
            )
        )  # fmt: skip
        return dd

    def window_bounds(self, pos, win_start, win_end):
        """Site-index ranges [lo, hi) of inclusive position windows; int32 device tensors."""
        torch = _torch()
        ws = torch.as_tensor(np.asarray(win_start, dtype=np.int64)).to(self.device)
        we = torch.as_tensor(np.asarray(win_end, dtype=np.int64)).to(self.device)
        n_w = int(ws.numel())
        lo = self._empty((n_w,), torch.int32)
        hi = self._empty((n_w,), torch.int32)
        _ffi.check(
            self.lib.sai_window_bounds(
                self.ctx, self._ptr(pos), int(pos.numel()), n_w, self._ptr(ws), self._ptr(we), self._ptr(lo),
                self._ptr(hi), self._stream(),
            )
        )  # fmt: skip
        return lo, hi

    def window_stats_async(self, tgt_freq, planes, sets, lo, hi, pos, bufs):
        """Enqueue the window kernel into caller-held buffers (no sync).  ``planes`` = the flag planes
        of exactly these sets (a column slice of a larger planes tensor is fine); ``bufs`` =
        (records u8 [S*W*24], offsets i64 [S*W*2], cdd_u i32, cdd_q i32, totals i64 [2], ...)."""
        n_sets, n_sites = len(sets), int(tgt_freq.numel())
        if n_sets > _ffi.SAI_MAX_SETS:
            raise ValueError("window_stats_async handles at most SAI_MAX_SETS sets per call")
        records, offsets, cdd_u, cdd_q, totals = bufs[:5]
        pl_ptr, pl_stride = self._planes_arg(planes, n_sets, n_sites)
        _ffi.check(
            self.lib.sai_window_stats(
                self.ctx, n_sites, self._ptr(tgt_freq), pl_ptr, pl_stride, n_sets, self._params_array(sets),
                int(lo.numel()), self._ptr(lo), self._ptr(hi), self._ptr(pos) if pos is not None else None,
                self._ptr(records), self._ptr(offsets), self._ptr(cdd_u), int(cdd_u.numel()), self._ptr(cdd_q),
                int(cdd_q.numel()), self._ptr(totals), self._stream(),
            )
        )  # fmt: skip

    def alloc_window_bufs(self, n_sets, n_windows, cap_u, cap_q, joined: bool = False):
        """(records, offsets, cdd_u, cdd_q, totals, head): records, offsets and totals are views of
        the one contiguous byte buffer ``head``, so a single copy brings them to the host.  ``joined``: the
        two candidate lists lie right behind ``head`` in the same allocation (a seventh entry is the whole of
        it): ONE copy then brings records and lists -- every copy of the runtime's starts ~12 us after the
        kernel before it has ended (profiles/history/r04_score_windows_call.txt)."""
        torch = _torch()
        n_rec = n_sets * n_windows
        rec_bytes, off_bytes = n_rec * RECORD_DTYPE.itemsize, n_rec * 16  # 24 B records keep 8-byte alignment
        tot_bytes = 8 * int(self.lib.sai_window_total_words(n_sets, n_windows))  # 2 totals + the scan's scratch
        head_bytes = rec_bytes + off_bytes + tot_bytes
        cap_u, cap_q = max(int(cap_u), 1), max(int(cap_q), 1)
        if joined:
            whole = self._empty((head_bytes + 4 * (cap_u + cap_q),), torch.uint8)
            head = whole[:head_bytes]
            cdd_u = whole[head_bytes : head_bytes + 4 * cap_u].view(torch.int32)
            cdd_q = whole[head_bytes + 4 * cap_u :].view(torch.int32)
        else:
            whole = None
            head = self._empty((head_bytes,), torch.uint8)
            cdd_u, cdd_q = self._empty((cap_u,), torch.int32), self._empty((cap_q,), torch.int32)
        bufs = (
            head[:rec_bytes],
            head[rec_bytes : rec_bytes + off_bytes].view(torch.int64),
            cdd_u,
            cdd_q,
            head[rec_bytes + off_bytes :].view(torch.int64),
            head,
        )
        return bufs + (whole,) if joined else bufs

    def window_stats(self, tgt_freq, planes, sets, lo, hi, pos=None, cap_hint=1 << 16) -> WindowResults:
        """Records and candidate lists of every (set, window), copied to the host."""
        n_sets, n_w = len(sets), int(lo.numel())
        rec_parts, off_parts, u_parts, q_parts = [], [], [], []
        base_u = base_q = 0
        for s0 in range(0, n_sets, _ffi.SAI_MAX_SETS):
            chunk = sets[s0 : s0 + _ffi.SAI_MAX_SETS]
            fl = planes[:, PLANES * s0 : PLANES * (s0 + len(chunk))]
            cap_u = cap_q = cap_hint
            while True:
                bufs = self.alloc_window_bufs(len(chunk), n_w, cap_u, cap_q)
                self.window_stats_async(tgt_freq, fl, chunk, lo, hi, pos, bufs)
                need_u, need_q = (int(v) for v in bufs[4][:2].cpu().tolist())
                if need_u <= bufs[2].numel() and need_q <= bufs[3].numel():
                    break
                cap_u, cap_q = max(need_u, 1), max(need_q, 1)
            rec = np.frombuffer(bufs[0].cpu().numpy().tobytes(), dtype=RECORD_DTYPE).reshape(len(chunk), n_w)
            off = bufs[1].cpu().numpy().reshape(len(chunk), n_w, 2).copy()
            off[:, :, 0] += base_u
            off[:, :, 1] += base_q
            rec_parts.append(rec)
            off_parts.append(off)
            u_parts.append(bufs[2][:need_u].cpu().numpy())
            q_parts.append(bufs[3][:need_q].cpu().numpy())
            base_u += need_u
            base_q += need_q
        return WindowResults(
            np.concatenate(rec_parts, axis=0),
            np.concatenate(off_parts, axis=0),
            np.concatenate(u_parts) if u_parts else np.zeros(0, np.int32),
            np.concatenate(q_parts) if q_parts else np.zeros(0, np.int32),
        )

    # -- measurement aid ------------------------------------------------------------------

    def probe_stream_read(self, buf, repeats: int = 5, launches: int = 4) -> float:
        """GB/s of the library's plain streaming-read kernel over ``buf``: best of ``repeats``
        timed regions of ``launches`` back-to-back launches each (so the queue stays full and
        the host's launch gap is not billed to the kernel)."""
        torch = _torch()
        out = self._empty((1,), torch.int32)
        n = (buf.numel() * buf.element_size()) & ~15
        best = 0.0
        for _ in range(repeats + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            out.zero_()
            e0.record()
            for _ in range(launches):
                _ffi.check(self.lib.sai_probe_stream_read(self.ctx, self._ptr(buf), n, self._ptr(out), self._stream()))
            e1.record()
            e1.synchronize()
            best = max(best, launches * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        return best

    # -- synthetic data --------------------------------------------------------------------

    def synth_population(self, seed, chrom, site0, n_sites, pop_stream, n_ind, ploidy=2, missing_per_million=0, out=None):
        """synth-v1 genotypes of sites [site0, site0 + n_sites) as a tiled block; ``out`` = an int8
        device tensor of exactly ``sai_tiled_bytes`` bytes to fill instead of a fresh allocation (a
        tile-aligned slice of a larger block that holds several pieces)."""
        torch = _torch()
        nbytes = self.lib.sai_tiled_bytes(n_sites, n_ind)
        if out is None:
            tiles = self._empty((nbytes,), torch.int8)
        else:
            if out.dtype != torch.int8 or out.numel() != nbytes or not out.is_contiguous():
                raise ValueError("out must be a contiguous int8 tensor of sai_tiled_bytes(n_sites, n_ind) bytes")
            tiles = out
        _ffi.check(
            self.lib.sai_synth_fill(
                self.ctx, seed, chrom, site0, n_sites, pop_stream, n_ind, ploidy, missing_per_million,
                self._ptr(tiles), self._stream(),
            )
        )  # fmt: skip
        return TiledPop(tiles, int(n_sites), int(n_ind))

    def synth_positions(self, seed, chrom, n_sites, site0=0):
        """int32 device positions of sites [site0, site0 + n_sites) (prefix sum of the gaps of
        sites 0..site0+n_sites-1, so any shard sees the same coordinates)."""
        torch = _torch()
        total = site0 + n_sites
        gaps = self._empty((total,), torch.int32)
        _ffi.check(self.lib.sai_synth_gaps(self.ctx, seed, chrom, 0, total, self._ptr(gaps), self._stream()))
        pos = torch.cumsum(gaps, dim=0, dtype=torch.int64)
        if total and int(pos[-1]) >= 2**31:
            raise ValueError("synthetic chromosome exceeds int32 coordinates")
        return pos[site0:].to(torch.int32).contiguous()
