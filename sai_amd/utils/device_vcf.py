"""VCF region -> dosages resident in HBM, tokenised on the GPU.

The host side (libsaihip's ``sai_vcf_stream_*``) reads / inflates the file and indexes its record
lines while the genotype text crosses PCIe as it is; ``sai_tokenize_gt`` turns the text into int8
dosages on the GPU.  The producer thread fills one pinned buffer while the other one is being copied
and tokenised, so reading, PCIe and the kernel overlap.  Same rules and same bytes as the host
tokenizer (``native_vcf.load_dosage``); a line the GPU flags is handed to the host reader, which
produces the reference's error text.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from .. import _ffi
from .native_vcf import default_threads, load_dosage

BUFFER_BYTES = 32 << 20
INFLATE_BATCH_BYTES = 128 << 20
_MEMBER_BYTES = 32  # sizeof(sai_bgzf_member)


class _Fallback(Exception):
    """The bgzip-on-the-GPU route cannot serve this read; the host-inflating stream takes it."""


def _io_error(lib) -> ValueError:
    return ValueError(lib.sai_last_error().decode("utf-8", "replace"))


def load_dosage_device(eng, vcf_file: str, chr_name: str, samples: Sequence[str], ploidies: Sequence[int],
                       start: Optional[int] = None, end: Optional[int] = None, anc_allele_file: Optional[str] = None,
                       n_threads: Optional[int] = None, buffer_bytes: Optional[int] = None):  # fmt: skip
    """(pos int32 host array [n], dosage int8 DEVICE tensor [n][len(samples)], n_matched,
    n_anc_entries) for one region -- ``load_dosage`` with the result left in HBM."""
    import torch

    lib = eng.lib
    cap = int(buffer_bytes or os.environ.get("SAI_AMD_INGEST_BUFFER", BUFFER_BYTES))
    if os.environ.get("SAI_AMD_GPU_INFLATE", "1") != "0":
        try:
            # 2 048 members are in flight on the chip at a time: a batch of 128 MiB of text fills it
            icap = int(buffer_bytes or os.environ.get("SAI_AMD_INFLATE_BATCH", INFLATE_BATCH_BYTES))
            got = _load_bgzf_device(eng, vcf_file, chr_name, samples, ploidies, start, end, anc_allele_file, n_threads, icap)
            if got is not None:
                return got
        except _Fallback:
            pass
    st = eng.__dict__.setdefault("_ingest_state", {})
    if st.get("cap") != cap:  # pinned staging + device text buffers, kept for the next call
        st.clear()
        st["cap"] = cap
        st["pinned"] = [torch.empty((cap,), dtype=torch.uint8).pin_memory() for _ in range(2)]
        st["text"] = [torch.empty((cap + 16,), dtype=torch.uint8, device=eng.device) for _ in range(2)]
        st["stream"] = torch.cuda.Stream(device=eng.device)
    pinned, text, side = st["pinned"], st["text"], st["stream"]
    n = len(samples)
    names = (C.c_char_p * n)(*[s.encode() for s in samples])
    pl = (C.c_int32 * n)(*[int(p) for p in ploidies])
    handle = C.c_void_p()
    if lib.sai_vcf_stream_open(
        os.fsencode(vcf_file), str(chr_name).encode(), -1 if start is None else int(start), -1 if end is None else int(end),
        n, names, pl, os.fsencode(anc_allele_file) if anc_allele_file else None, n_threads or default_threads(),
        C.c_void_p(pinned[0].data_ptr()), C.c_void_p(pinned[1].data_ptr()), cap, C.byref(handle),
    ):  # fmt: skip
        raise _io_error(lib)
    ploidy_dev = torch.tensor([int(p) for p in ploidies], dtype=torch.int32, device=eng.device)
    slot_dev, n_cols = None, 0
    outs, stats, pos_parts, copied = [], [], [], []
    try:
        buf, n_text, n_lines, done = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int32()
        p_off, p_len, p_pos, p_flip, p_gi = (C.c_void_p() for _ in range(5))
        while True:
            if copied:
                copied[-1].synchronize()  # the H2D copy of the previous batch has left its pinned buffer
            if lib.sai_vcf_stream_next(handle, C.byref(buf), C.byref(n_text), C.byref(n_lines), C.byref(p_off), C.byref(p_len),
                                       C.byref(p_pos), C.byref(p_flip), C.byref(p_gi), C.byref(done)):  # fmt: skip
                raise _io_error(lib)
            if done.value:
                break
            nl, nb, b = int(n_lines.value), int(n_text.value), int(buf.value)
            if slot_dev is None:
                cols = C.c_int32()
                if lib.sai_vcf_stream_selection(handle, None, 0, C.byref(cols), None, None):
                    raise _io_error(lib)
                n_cols = int(cols.value)
                slots = np.empty(max(n_cols, 1), dtype=np.int32)
                if lib.sai_vcf_stream_selection(handle, slots.ctypes.data_as(C.c_void_p), n_cols, C.byref(cols), None, None):
                    raise _io_error(lib)
                slot_dev = torch.from_numpy(slots[:n_cols].copy()).to(eng.device)

            def arr(ptr, ctype, dtype):
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(nl,)).astype(dtype, copy=True)

            with torch.cuda.stream(side):
                text[b][: nb].copy_(pinned[b][:nb], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
                copied.append(ev)
                if nl == 0:
                    continue
                pos_parts.append(arr(p_pos, C.c_int32, np.int32))
                d_off = torch.from_numpy(arr(p_off, C.c_int64, np.int64)).to(eng.device, non_blocking=True)
                d_len = torch.from_numpy(arr(p_len, C.c_int32, np.int32)).to(eng.device, non_blocking=True)
                d_flip = torch.from_numpy(arr(p_flip, C.c_uint8, np.uint8)).to(eng.device, non_blocking=True)
                d_gi = torch.from_numpy(arr(p_gi, C.c_uint8, np.uint8)).to(eng.device, non_blocking=True)
                out = torch.empty((nl, n), dtype=torch.int8, device=eng.device)
                status = torch.empty((nl,), dtype=torch.int32, device=eng.device)
                _ffi.check(
                    lib.sai_tokenize_gt(eng.ctx, C.c_void_p(text[b].data_ptr()), (nb + 3) & ~3, nl, eng._ptr(d_off), eng._ptr(d_len),
                                        eng._ptr(d_flip), eng._ptr(d_gi), n_cols, eng._ptr(slot_dev), n, eng._ptr(ploidy_dev),
                                        eng._ptr(out), eng._ptr(status), C.c_void_p(side.cuda_stream))
                )  # fmt: skip
                outs.append(out)
                stats.append(status)
        n_match, n_anc, cols = C.c_int64(), C.c_int64(), C.c_int32()
        have_header = lib.sai_vcf_stream_selection(handle, None, 0, C.byref(cols), C.byref(n_match), C.byref(n_anc)) == 0
    finally:
        lib.sai_vcf_stream_close(handle)
        side.synchronize()  # also on an error: the staging buffers are reused by the next call
    if stats and bool(torch.cat(stats).any()):
        # a line the host reader refuses: let it say why, in the reference's words
        load_dosage(vcf_file, chr_name, samples, ploidies, start, end, anc_allele_file, n_threads)
        raise ValueError(f"{vcf_file}: the GPU tokenizer flagged a line the host reader accepts")
    torch.cuda.current_stream(eng.device).wait_stream(side)
    pos = np.concatenate(pos_parts) if pos_parts else np.zeros(0, dtype=np.int32)
    dos = torch.cat(outs) if len(outs) > 1 else (outs[0] if outs else torch.empty((0, n), dtype=torch.int8, device=eng.device))
    return pos, dos, (int(n_match.value) if have_header else 0), (int(n_anc.value) if have_header else 0)


def _load_bgzf_device(eng, vcf_file, chr_name, samples, ploidies, start, end, anc_allele_file, n_threads, cap):
    """``load_dosage_device`` for a bgzip file without a region seek: the compressed members cross
    PCIe and are inflated by ``sai_inflate_bgzf`` (one wavefront per member); a second
    launch checks every member's CRC-32; the text comes back to the host ONCE, for
    ``sai_vcf_index_text`` (header, record index -- the same code the host stream runs), and is
    tokenised where it lies in HBM.  Batch k+1 is being inflated
    while the host indexes batch k.  Returns None when the file is not bgzip (or a tabix index
    serves the region): the caller falls back to the host-inflating stream."""
    import torch

    lib = eng.lib
    room = max(1 << 20, min(cap // 4, 8 << 20))  # the incomplete last line of a batch is carried in front of the next one
    comp_cap = cap // 4 + (1 << 20)
    st = eng.__dict__.setdefault("_inflate_state", {})
    if st.get("cap") != cap:
        st.clear()
        st["cap"] = cap
        st["comp_host"] = [torch.empty((comp_cap,), dtype=torch.uint8).pin_memory() for _ in range(2)]
        st["comp_dev"] = [torch.empty((comp_cap,), dtype=torch.uint8, device=eng.device) for _ in range(2)]
        st["text_host"] = [torch.empty((room + cap + 16,), dtype=torch.uint8).pin_memory() for _ in range(2)]
        st["text_dev"] = [torch.empty((room + cap + 16,), dtype=torch.uint8, device=eng.device) for _ in range(2)]
        st["flag_host"] = [torch.zeros((1,), dtype=torch.int32).pin_memory() for _ in range(2)]
        st["side"] = torch.cuda.Stream(device=eng.device)
        st["copy"] = torch.cuda.Stream(device=eng.device)
        st["d2h"] = torch.cuda.Stream(device=eng.device)
    comp_host, comp_dev, text_host, text_dev = st["comp_host"], st["comp_dev"], st["text_host"], st["text_dev"]
    flag_host, side, copy, d2h = st["flag_host"], st["side"], st["copy"], st["d2h"]
    n = len(samples)
    names = (C.c_char_p * n)(*[s.encode() for s in samples])
    pl = (C.c_int32 * n)(*[int(p) for p in ploidies])
    handle = C.c_void_p()
    rc = lib.sai_bgzf_stream_open(
        os.fsencode(vcf_file), str(chr_name).encode(), -1 if start is None else int(start), -1 if end is None else int(end),
        n, names, pl, os.fsencode(anc_allele_file) if anc_allele_file else None, n_threads or default_threads(),
        C.c_void_p(comp_host[0].data_ptr()), C.c_void_p(comp_host[1].data_ptr()), comp_cap, cap, C.byref(handle),
    )  # fmt: skip
    if rc == _ffi.SAI_ERR_UNSUPPORTED:
        return None
    if rc:
        raise _io_error(lib)
    ploidy_dev = torch.tensor([int(p) for p in ploidies], dtype=torch.int32, device=eng.device)
    state = {"slot_dev": None, "n_cols": 0}
    outs, stats, pos_parts = [], [], []
    usable, n_lines, idone = C.c_int64(), C.c_int64(), C.c_int32()
    p_off, p_len, p_pos, p_flip, p_gi = (C.c_void_p() for _ in range(5))

    def index_and_tokenize(b, base, n_bytes, n_carry, table, is_last):
        """Index text_host[b][base : base + n_bytes] and tokenise its record lines from text_dev[b]."""
        tab_ptr = C.c_void_p(table.ctypes.data) if table is not None else None
        if lib.sai_vcf_index_text(handle, C.c_void_p(text_host[b].data_ptr() + base), n_bytes, n_carry, tab_ptr,
                                  0 if table is None else len(table) // _MEMBER_BYTES, 1 if is_last else 0, C.byref(usable),
                                  C.byref(n_lines), C.byref(p_off), C.byref(p_len), C.byref(p_pos), C.byref(p_flip), C.byref(p_gi),
                                  C.byref(idone)):  # fmt: skip
            raise _io_error(lib)
        nl = int(n_lines.value)
        if nl == 0:
            return
        if state["slot_dev"] is None:
            cols = C.c_int32()
            if lib.sai_bgzf_stream_selection(handle, None, 0, C.byref(cols), None, None):
                raise _io_error(lib)
            state["n_cols"] = int(cols.value)
            slots = np.empty(max(state["n_cols"], 1), dtype=np.int32)
            if lib.sai_bgzf_stream_selection(handle, slots.ctypes.data_as(C.c_void_p), state["n_cols"], C.byref(cols), None, None):
                raise _io_error(lib)
            state["slot_dev"] = torch.from_numpy(slots[: state["n_cols"]].copy()).to(eng.device)

        def arr(ptr, ctype, dtype):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(nl,)).astype(dtype, copy=True)

        pos_parts.append(arr(p_pos, C.c_int32, np.int32))
        with torch.cuda.stream(side):
            d_off = torch.from_numpy(arr(p_off, C.c_int64, np.int64) + base).to(eng.device, non_blocking=True)
            d_len = torch.from_numpy(arr(p_len, C.c_int32, np.int32)).to(eng.device, non_blocking=True)
            d_flip = torch.from_numpy(arr(p_flip, C.c_uint8, np.uint8)).to(eng.device, non_blocking=True)
            d_gi = torch.from_numpy(arr(p_gi, C.c_uint8, np.uint8)).to(eng.device, non_blocking=True)
            out = torch.empty((nl, n), dtype=torch.int8, device=eng.device)
            status = torch.empty((nl,), dtype=torch.int32, device=eng.device)
            _ffi.check(
                lib.sai_tokenize_gt(eng.ctx, C.c_void_p(text_dev[b].data_ptr()), (base + n_bytes + 3) & ~3, nl, eng._ptr(d_off),
                                    eng._ptr(d_len), eng._ptr(d_flip), eng._ptr(d_gi), state["n_cols"], eng._ptr(state["slot_dev"]), n,
                                    eng._ptr(ploidy_dev), eng._ptr(out), eng._ptr(status), C.c_void_p(side.cuda_stream))
            )  # fmt: skip
            outs.append(out)
            stats.append(status)

    import time as _time

    trace = {} if os.environ.get("SAI_AMD_INGEST_TRACE") else None
    t_mark = [_time.perf_counter()]

    def lap(name):
        if trace is not None:
            now = _time.perf_counter()
            trace[name] = trace.get(name, 0.0) + now - t_mark[0]
            t_mark[0] = now

    try:
        buf, n_comp, n_mem, n_text, done = C.c_int32(), C.c_int64(), C.c_int32(), C.c_int64(), C.c_int32()
        table_p = C.c_void_p()
        prev, h2d, reader_done = None, None, False
        # Per batch: H2D of the compressed bytes (copy stream) -> inflate (side stream) -> D2H of the
        # text (d2h stream), all enqueued as soon as the reader hands the batch over; the host then
        # indexes the batch BEFORE it while those run.  The incomplete last line of batch k is copied
        # in front of batch k+1 on both sides (host memcpy, D2D copy) once the index of k is known.
        while True:
            batch = None
            if not reader_done:
                lap("other")
                if h2d is not None:
                    h2d.synchronize()  # next() releases the pinned buffer of the batch before
                lap("wait_h2d")
                if lib.sai_bgzf_stream_next(handle, C.byref(buf), C.byref(n_comp), C.byref(n_mem), C.byref(table_p),
                                            C.byref(n_text), C.byref(done)):  # fmt: skip
                    raise _io_error(lib)
                lap("wait_reader")
                if done.value:
                    reader_done = True
                else:
                    b, nc, nm, nt = int(buf.value), int(n_comp.value), int(n_mem.value), int(n_text.value)
                    table = np.ctypeslib.as_array(C.cast(table_p, C.POINTER(C.c_uint8)), shape=(nm * _MEMBER_BYTES,)).copy()
                    with torch.cuda.stream(copy):
                        comp_dev[b][:nc].copy_(comp_host[b][:nc], non_blocking=True)
                        h2d = torch.cuda.Event()
                        h2d.record(copy)
                    with torch.cuda.stream(side):
                        side.wait_event(h2d)
                        d_tab = torch.from_numpy(table).to(eng.device, non_blocking=True)
                        d_stat = torch.empty((nm,), dtype=torch.int32, device=eng.device)
                        _ffi.check(
                            lib.sai_inflate_bgzf(eng.ctx, C.c_void_p(comp_dev[b].data_ptr()), nc, C.c_void_p(d_tab.data_ptr()), nm,
                                                 C.c_void_p(text_dev[b].data_ptr() + room), nt, C.c_void_p(d_stat.data_ptr()),
                                                 C.c_void_p(side.cuda_stream))
                        )  # fmt: skip
                        inflated = torch.cuda.Event()
                        inflated.record(side)
                    with torch.cuda.stream(d2h):
                        d2h.wait_event(inflated)
                        text_host[b][room : room + nt].copy_(text_dev[b][room : room + nt], non_blocking=True)
                        flag_host[b].copy_((d_stat != 0).sum(dtype=torch.int32).reshape(1), non_blocking=True)
                        back = torch.cuda.Event()
                        back.record(d2h)
                    batch = {"b": b, "n_text": nt, "table": table, "d_stat": d_stat, "d_tab": d_tab, "d2h": back, "carry": 0}
            lap("enqueue_inflate")
            if prev is not None:
                prev["d2h"].synchronize()
                lap("wait_d2h")
                if int(flag_host[prev["b"]][0]):
                    raise ValueError(f"{vcf_file}: BGZF block fails to inflate or its CRC")
                pb, base = prev["b"], room - prev["carry"]
                index_and_tokenize(pb, base, prev["carry"] + prev["n_text"], prev["carry"], None, False)  # CRCs: checked on the GPU
                lap("index_and_tokenize")
                left = prev["carry"] + prev["n_text"] - int(usable.value)
                at = base + int(usable.value)
                if left > room:
                    raise _Fallback  # a record line longer than the carry room
                if idone.value:
                    break
                if batch is None:
                    if left:  # the file ends without a newline
                        index_and_tokenize(pb, at, left, left, None, True)
                    break
                if left:
                    nb = batch["b"]
                    text_host[nb][room - left : room].copy_(text_host[pb][at : at + left])
                    with torch.cuda.stream(side):
                        text_dev[nb][room - left : room].copy_(text_dev[pb][at : at + left], non_blocking=True)
                    batch["carry"] = left
            elif batch is None:  # not a single batch: let the indexer say what is missing
                index_and_tokenize(0, room, 0, 0, None, True)
                break
            prev = batch
        n_match, n_anc, cols = C.c_int64(), C.c_int64(), C.c_int32()
        have_header = lib.sai_bgzf_stream_selection(handle, None, 0, C.byref(cols), C.byref(n_match), C.byref(n_anc)) == 0
    finally:
        lap("other")
        lib.sai_bgzf_stream_close(handle)
        side.synchronize()  # also on an error: the staging buffers are reused by the next call
        copy.synchronize()
        d2h.synchronize()
        lap("close_and_drain")
    if stats and bool(torch.cat(stats).any()):
        load_dosage(vcf_file, chr_name, samples, ploidies, start, end, anc_allele_file, n_threads)
        raise ValueError(f"{vcf_file}: the GPU tokenizer flagged a line the host reader accepts")
    torch.cuda.current_stream(eng.device).wait_stream(side)
    pos = np.concatenate(pos_parts) if pos_parts else np.zeros(0, dtype=np.int32)
    dos = torch.cat(outs) if len(outs) > 1 else (outs[0] if outs else torch.empty((0, n), dtype=torch.int8, device=eng.device))
    lap("status_and_concat")
    if trace is not None:
        print("bgzf route, ms:", " ".join(f"{k}={1e3 * v:.1f}" for k, v in trace.items()), flush=True)
    return pos, dos, (int(n_match.value) if have_header else 0), (int(n_anc.value) if have_header else 0)
