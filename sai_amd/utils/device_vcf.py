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
INFLATE_BATCH_BYTES = 3968 << 16  # 248 MiB: at most 3 968 full members, under the 4 096 the chip holds at a time
_MEMBER_BYTES = 32  # sizeof(sai_bgzf_member)


def _inflate_batch_for(vcf_file) -> int:
    """Text bytes per GPU-inflate batch: the full 248 MiB for files that can fill it, a quarter-step
    size class for small ones (the staging buffers are pinned and kept per size: a 100 kB file should
    not page-lock 400 MB).  Only a ceiling: a file that inflates to more simply takes more batches."""
    try:
        guess = os.path.getsize(vcf_file) * 16  # genotype text compresses 10-20x
    except OSError:
        return INFLATE_BATCH_BYTES
    cap = 1 << 20
    while cap < guess and cap < INFLATE_BATCH_BYTES:
        cap <<= 2
    return min(cap, INFLATE_BATCH_BYTES)


class _Fallback(Exception):
    """The bgzip-on-the-GPU route cannot serve this read; the host-inflating stream takes it."""


class _TextIndex(Exception):
    """The line heads would be most of the text (short lines) or do not reach the ninth tab: index
    this file from the whole text instead."""


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
            # 4 096 members are in flight on the chip at a time: a batch of 248 MiB of text fills it in one round
            icap = int(buffer_bytes or os.environ.get("SAI_AMD_INFLATE_BATCH", 0)) or _inflate_batch_for(vcf_file)
            try:
                got = _load_bgzf_device(eng, vcf_file, chr_name, samples, ploidies, start, end, anc_allele_file, n_threads, icap)
            except _TextIndex:
                got = _load_bgzf_device(eng, vcf_file, chr_name, samples, ploidies, start, end, anc_allele_file, n_threads, icap,
                                        index_from="text")  # fmt: skip
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


def _load_bgzf_device(eng, vcf_file, chr_name, samples, ploidies, start, end, anc_allele_file, n_threads, cap, index_from=None,
                      positions_only=False):  # fmt: skip
    """``load_dosage_device`` for a bgzip file.  A region of a file with a usable ``.tbi`` is a seek: the
    reader hands over only the members that hold the region (``sai_bgzf_stream_region``), and the line
    table of the first batch starts behind the text that precedes the region's first record.  The compressed members cross
    PCIe and are inflated by ``sai_inflate_bgzf`` (one wavefront per member; a second launch checks
    every member's CRC-32).  The record index -- chromosome / region filter, POS, the ancestral-allele
    decision, where the sample columns start: the host's ``index_lines`` rules -- is made from the
    line table the GPU extracts (``sai_text_line_starts`` / ``_heads``: line offsets + the fixed
    columns of every line, a few MB) or, ``index_from="text"``, from the whole text copied back once;
    the text is tokenised where it lies in HBM either way.  Batch k+1 is inflated and scanned while
    the host indexes batch k.  Returns None when the file is not bgzip: the caller falls back to the
    host-inflating stream.  ``positions_only`` (no samples):
    the record index alone, nothing is tokenised."""
    import torch

    lib = eng.lib
    index_from = index_from or os.environ.get("SAI_AMD_BGZF_INDEX", "heads")
    room = max(1 << 20, min(cap // 4, 8 << 20))  # the incomplete last line of a batch is carried in front of the next one
    comp_cap = cap // 4 + (1 << 20)
    line_cap = (room + cap) // 48 + 16
    heads_cap = (room + cap) // 4
    st = eng.__dict__.setdefault("_inflate_state", {})
    if st.get("cap") != cap:
        st.clear()
        st["cap"] = cap
        st["comp_host"] = [torch.empty((comp_cap,), dtype=torch.uint8).pin_memory() for _ in range(2)]
        st["comp_dev"] = [torch.empty((comp_cap,), dtype=torch.uint8, device=eng.device) for _ in range(2)]
        st["text_dev"] = [torch.empty((room + cap + 32,), dtype=torch.uint8, device=eng.device) for _ in range(2)]
        st["flag_host"] = [torch.zeros((4,), dtype=torch.int32).pin_memory() for _ in range(2)]
        st["side"] = torch.cuda.Stream(device=eng.device)
        st["copy"] = torch.cuda.Stream(device=eng.device)
        st["d2h"] = torch.cuda.Stream(device=eng.device)
        st["tok"] = torch.cuda.Stream(device=eng.device)
    if index_from == "text" and "text_host" not in st:
        st["text_host"] = [torch.empty((room + cap + 32,), dtype=torch.uint8).pin_memory() for _ in range(2)]
    if index_from != "text" and "starts_dev" not in st:
        st["starts_dev"] = [torch.empty((line_cap + 1,), dtype=torch.int64, device=eng.device) for _ in range(2)]
        st["info_dev"] = [torch.empty((line_cap,), dtype=torch.int32, device=eng.device) for _ in range(2)]
        st["heads_dev"] = [torch.empty((heads_cap,), dtype=torch.uint8, device=eng.device) for _ in range(2)]
        st["scratch_dev"] = torch.empty(((room + cap + 32) // 4096 + 4,), dtype=torch.int32, device=eng.device)
        st["info4_dev"] = [torch.zeros((4,), dtype=torch.int32, device=eng.device) for _ in range(2)]
        # the pinned mirrors grow with the line counts actually seen (a 2 002-sample file has 31 000 lines
        # in a 248 MiB batch: 2 MB of table; page-locking for the worst case would cost 270 MB and 0.2 s)
        st["starts_host"] = [torch.empty((1,), dtype=torch.int64).pin_memory() for _ in range(2)]
        st["info_host"] = [torch.empty((1,), dtype=torch.int32).pin_memory() for _ in range(2)]
        st["heads_host"] = [torch.empty((1,), dtype=torch.uint8).pin_memory() for _ in range(2)]
        st["tail_host"] = torch.empty((1,), dtype=torch.uint8).pin_memory()
    comp_host, comp_dev, text_dev = st["comp_host"], st["comp_dev"], st["text_dev"]
    flag_host, side, copy, d2h, tok = st["flag_host"], st["side"], st["copy"], st["d2h"], st["tok"]
    tok_done = [None, None]  # per ring slot: the tokenizer that read text_dev[slot] last
    text_host = st.get("text_host")
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
    f_begin, f_stop, f_skip = C.c_int64(), C.c_int64(), C.c_int64()
    if lib.sai_bgzf_stream_region(handle, C.byref(f_begin), C.byref(f_stop), C.byref(f_skip)):
        lib.sai_bgzf_stream_close(handle)
        raise _io_error(lib)
    # text of the first member that belongs to records before the region (a tabix seek lands inside a member)
    state = {"slot_dev": None, "n_cols": 0, "skip": int(f_skip.value)}
    # what the last read took from the file (tests and tools/bgzf_rate.py look at it)
    st["last"] = {"file_begin": int(f_begin.value), "file_stop": int(f_stop.value), "first_text_skip": int(f_skip.value),
                  "members": 0, "comp_bytes": 0, "text_bytes": 0}  # fmt: skip
    outs, stats, pos_parts = [], [], []
    usable, n_lines, idone = C.c_int64(), C.c_int64(), C.c_int32()
    p_off, p_len, p_pos, p_flip, p_gi = (C.c_void_p() for _ in range(5))
    index_out = (C.byref(n_lines), C.byref(p_off), C.byref(p_len), C.byref(p_pos), C.byref(p_flip), C.byref(p_gi), C.byref(idone))

    def tokenize(b, base, n_bytes, ready=None):
        """The record lines the index call just reported (offsets relative to text_dev[b][base]).  With
        ``ready`` (the event behind the batch's text) the kernel runs on its own stream, beside the
        inflate of the next batch -- that one occupies a fifth of the wavefront slots."""
        nl = int(n_lines.value)
        if nl == 0:
            return
        if state["slot_dev"] is None and not positions_only:
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
        if positions_only:
            return
        stream = side if ready is None else tok
        with torch.cuda.stream(stream):
            if ready is not None:
                tok.wait_event(ready)
            d_off = torch.from_numpy(arr(p_off, C.c_int64, np.int64) + base).to(eng.device, non_blocking=True)
            d_len = torch.from_numpy(arr(p_len, C.c_int32, np.int32)).to(eng.device, non_blocking=True)
            d_flip = torch.from_numpy(arr(p_flip, C.c_uint8, np.uint8)).to(eng.device, non_blocking=True)
            d_gi = torch.from_numpy(arr(p_gi, C.c_uint8, np.uint8)).to(eng.device, non_blocking=True)
            out = torch.empty((nl, n), dtype=torch.int8, device=eng.device)
            status = torch.empty((nl,), dtype=torch.int32, device=eng.device)
            _ffi.check(
                lib.sai_tokenize_gt(eng.ctx, C.c_void_p(text_dev[b].data_ptr()), (base + n_bytes + 3) & ~3, nl, eng._ptr(d_off),
                                    eng._ptr(d_len), eng._ptr(d_flip), eng._ptr(d_gi), state["n_cols"], eng._ptr(state["slot_dev"]), n,
                                    eng._ptr(ploidy_dev), eng._ptr(out), eng._ptr(status), C.c_void_p(stream.cuda_stream))
            )  # fmt: skip
            outs.append(out)
            stats.append(status)
            if ready is not None:
                tok_done[b] = torch.cuda.Event()
                tok_done[b].record(tok)

    def index_text_and_tokenize(host_ptr, b, base, n_bytes, n_carry, is_last):
        if lib.sai_vcf_index_text(handle, C.c_void_p(host_ptr), n_bytes, n_carry, None, 0, 1 if is_last else 0, C.byref(usable),
                                  *index_out):  # fmt: skip
            raise _io_error(lib)
        tokenize(b, base, n_bytes)

    import time as _time

    trace = {} if os.environ.get("SAI_AMD_INGEST_TRACE") else None
    t_mark = [_time.perf_counter()]

    def lap(name):
        if trace is not None:
            now = _time.perf_counter()
            trace[name] = trace.get(name, 0.0) + now - t_mark[0]
            t_mark[0] = now

    last_inflate = [None, None]  # per ring slot: the event behind the inflate that read comp_dev[slot] last

    def fetch():
        """The next batch of members from the reader; its compressed bytes start for HBM (copy stream)."""
        buf, n_comp, n_mem, n_text, done = C.c_int32(), C.c_int64(), C.c_int32(), C.c_int64(), C.c_int32()
        table_p = C.c_void_p()
        if lib.sai_bgzf_stream_next(handle, C.byref(buf), C.byref(n_comp), C.byref(n_mem), C.byref(table_p), C.byref(n_text),
                                    C.byref(done)):  # fmt: skip
            raise _io_error(lib)
        lap("wait_reader")
        if done.value:
            return None
        b, nc, nm, nt = int(buf.value), int(n_comp.value), int(n_mem.value), int(n_text.value)
        st["last"]["members"] += nm
        st["last"]["comp_bytes"] += nc
        st["last"]["text_bytes"] += nt
        table = np.ctypeslib.as_array(C.cast(table_p, C.POINTER(C.c_uint8)), shape=(nm * _MEMBER_BYTES,)).copy()
        with torch.cuda.stream(copy):
            if last_inflate[b] is not None:
                copy.wait_event(last_inflate[b])  # the inflate two batches back may still be reading comp_dev[b]
            comp_dev[b][:nc].copy_(comp_host[b][:nc], non_blocking=True)
            h2d = torch.cuda.Event()
            h2d.record(copy)
        lap("copy_table")
        # "carry" = bytes in front of the batch's own text that belong to it (the incomplete last line of the
        # batch before); negative for the first batch of a region: that many bytes of its text are not its own
        carry, state["skip"] = -state["skip"], 0
        if -carry > nt:
            raise ValueError(f"{vcf_file}: the tabix index points behind the end of a BGZF block")
        return {"b": b, "n_comp": nc, "n_members": nm, "n_text": nt, "table": table, "h2d": h2d, "carry": carry}

    def launch(batch):
        """Inflate + CRC of a fetched batch.  Its H2D copy was started one batch earlier -- a copy issued
        while the inflate kernel holds the chip waits for it (measured: 2 ms instead of 0.2) -- so the
        wait here is short, and the reader gets its pinned buffer back."""
        b, nc, nm, nt = batch["b"], batch["n_comp"], batch["n_members"], batch["n_text"]
        batch["h2d"].synchronize()
        lib.sai_bgzf_stream_release(handle)
        lap("h2d_sync")
        with torch.cuda.stream(side):
            if tok_done[b] is not None:
                side.wait_event(tok_done[b])  # the tokenizer two batches back read text_dev[b]
            d_tab = torch.from_numpy(batch["table"]).to(eng.device, non_blocking=True)
            d_stat = torch.empty((nm,), dtype=torch.int32, device=eng.device)
            _ffi.check(
                lib.sai_inflate_bgzf(eng.ctx, C.c_void_p(comp_dev[b].data_ptr()), nc, C.c_void_p(d_tab.data_ptr()), nm,
                                     C.c_void_p(text_dev[b].data_ptr() + room), nt, C.c_void_p(d_stat.data_ptr()),
                                     C.c_void_p(side.cuda_stream))
            )  # fmt: skip
            batch["bad"] = (d_stat != 0).sum(dtype=torch.int32).reshape(1)
            batch["inflated"] = torch.cuda.Event()
            batch["inflated"].record(side)
            last_inflate[b] = batch["inflated"]
            batch["keep"] = (d_tab, d_stat)
        lap("enqueue_inflate")

    upcoming = [None, False]  # the batch fetched ahead, whether the reader has said "done"

    def next_batch():
        if not upcoming[1] and upcoming[0] is None:
            upcoming[0] = fetch()
            upcoming[1] = upcoming[0] is None
        batch = upcoming[0]
        if batch is None:
            return None
        launch(batch)
        upcoming[0] = None
        if not upcoming[1]:
            upcoming[0] = fetch()  # its H2D runs under the inflate just launched
            upcoming[1] = upcoming[0] is None
        return batch

    def move_carry(prev_b, at, left, batch):
        """The incomplete last line of the batch before, in front of `batch` on the device."""
        if left > room:
            raise _Fallback  # a record line longer than the carry room
        if left and batch is not None:
            with torch.cuda.stream(side):
                text_dev[batch["b"]][room - left : room].copy_(text_dev[prev_b][at : at + left], non_blocking=True)
            batch["carry"] = left

    try:
        if index_from == "text":
            # the whole text comes back once (d2h stream) and sai_vcf_index_text reads it
            prev = None
            while True:
                batch = next_batch()
                if batch is not None:
                    b, nt = batch["b"], batch["n_text"]
                    with torch.cuda.stream(d2h):
                        d2h.wait_event(batch["inflated"])
                        text_host[b][room : room + nt].copy_(text_dev[b][room : room + nt], non_blocking=True)
                        flag_host[b][:1].copy_(batch["bad"], non_blocking=True)
                        batch["back"] = torch.cuda.Event()
                        batch["back"].record(d2h)
                if prev is not None:
                    prev["back"].synchronize()
                    lap("wait_d2h")
                    if int(flag_host[prev["b"]][0]):
                        raise ValueError(f"{vcf_file}: BGZF block fails to inflate or its CRC")
                    pb, base, total = prev["b"], room - prev["carry"], prev["carry"] + prev["n_text"]
                    index_text_and_tokenize(text_host[pb].data_ptr() + base, pb, base, total, max(prev["carry"], 0), False)
                    lap("index_and_tokenize")
                    left, at = total - int(usable.value), base + int(usable.value)
                    if idone.value:
                        break
                    if batch is None:
                        if left:  # the file ends without a newline
                            index_text_and_tokenize(text_host[pb].data_ptr() + at, pb, at, left, left, True)
                        break
                    if left:
                        text_host[batch["b"]][room - left : room].copy_(text_host[pb][at : at + left])
                    move_carry(pb, at, left, batch)
                elif batch is None:
                    break
                prev = batch
        else:
            # the GPU finds the lines; the host indexes from their heads
            starts_dev, info_dev, heads_dev, scratch = st["starts_dev"], st["info_dev"], st["heads_dev"], st["scratch_dev"]
            info4_dev, starts_host, info_host, heads_host = st["info4_dev"], st["starts_host"], st["info_host"], st["heads_host"]
            tail_host = st["tail_host"]

            def scan_lines(batch):
                b, base, total = batch["b"], room - batch["carry"], batch["carry"] + batch["n_text"]
                with torch.cuda.stream(side):
                    _ffi.check(
                        lib.sai_text_line_starts(eng.ctx, C.c_void_p(text_dev[b].data_ptr() + base), total, line_cap,
                                                 C.c_void_p(starts_dev[b].data_ptr()), C.c_void_p(info_dev[b].data_ptr()),
                                                 C.c_void_p(scratch.data_ptr()), C.c_void_p(info4_dev[b].data_ptr()),
                                                 C.c_void_p(side.cuda_stream))
                    )  # fmt: skip
                    flag_host[b][:3].copy_(info4_dev[b][:3], non_blocking=True)
                    flag_host[b][3:].copy_(batch["bad"], non_blocking=True)
                    batch["scanned"] = torch.cuda.Event()
                    batch["scanned"].record(side)

            def fetch_table(batch):
                """Wait for the scan, gather the heads and start their copy to the host."""
                b, base, total = batch["b"], room - batch["carry"], batch["carry"] + batch["n_text"]
                batch["scanned"].synchronize()
                lap("wait_scan")
                n_l, fixed, overflow, bad = (int(v) for v in flag_host[b].tolist())
                if bad:
                    raise ValueError(f"{vcf_file}: BGZF block fails to inflate or its CRC")
                hb = max(16, (fixed + 3) & ~3)
                if overflow or fixed > 4096 or n_l * hb > heads_cap:
                    raise _TextIndex  # short lines / far fixed columns: the whole-text index serves this file
                batch["n_lines"], batch["hb"] = n_l, hb
                for buf, need, dt in ((starts_host, n_l + 1, torch.int64), (info_host, max(n_l, 1), torch.int32),
                                      (heads_host, max(n_l * hb, 1), torch.uint8)):
                    if buf[b].numel() < need:
                        buf[b] = torch.empty((1 << (need - 1).bit_length(),), dtype=dt).pin_memory()
                with torch.cuda.stream(d2h):  # its own stream: the next batch's inflate is already queued on `side`
                    _ffi.check(
                        lib.sai_text_line_heads(eng.ctx, C.c_void_p(text_dev[b].data_ptr() + base), total,
                                                C.c_void_p(starts_dev[b].data_ptr()), n_l, hb, C.c_void_p(heads_dev[b].data_ptr()),
                                                C.c_void_p(d2h.cuda_stream))
                    )  # fmt: skip
                    starts_host[b][: n_l + 1].copy_(starts_dev[b][: n_l + 1], non_blocking=True)
                    info_host[b][: max(n_l, 1)].copy_(info_dev[b][: max(n_l, 1)], non_blocking=True)
                    heads_host[b][: max(n_l * hb, 1)].copy_(heads_dev[b][: max(n_l * hb, 1)], non_blocking=True)
                    batch["tabled"] = torch.cuda.Event()
                    batch["tabled"].record(d2h)

            def index_batch(batch):
                b, base, total = batch["b"], room - batch["carry"], batch["carry"] + batch["n_text"]
                batch["tabled"].synchronize()
                lap("wait_table")
                n_l = batch["n_lines"]
                if lib.sai_vcf_index_heads(handle, C.c_void_p(heads_host[b].data_ptr()), batch["hb"], C.c_void_p(starts_host[b].data_ptr()),
                                           C.c_void_p(info_host[b].data_ptr()), n_l, *index_out):  # fmt: skip
                    raise _io_error(lib)
                tokenize(b, base, total, ready=batch["scanned"])
                lap("index_and_tokenize")
                used = int(starts_host[b][n_l]) if n_l else 0
                return base + used, total - used  # where the incomplete last line lies, and its length

            prev = None
            while True:
                batch = next_batch()
                at = left = 0
                if prev is not None:
                    # the line table of `prev` is on its way; its last offset says what is carried over
                    prev["scanned"].synchronize()
                    fetch_table(prev)
                    prev["tabled"].synchronize()
                    n_l = prev["n_lines"]
                    used = int(starts_host[prev["b"]][n_l]) if n_l else 0
                    at, left = room - prev["carry"] + used, prev["carry"] + prev["n_text"] - used
                    move_carry(prev["b"], at, left, batch)
                if batch is not None:
                    scan_lines(batch)  # runs behind the inflate of `batch`, while the host indexes `prev`
                if prev is not None:
                    index_batch(prev)
                    if idone.value:
                        break
                    if batch is None:
                        if left:  # the file ends without a newline: those few bytes come to the host as text
                            if tail_host.numel() < left:
                                tail_host = st["tail_host"] = torch.empty((1 << (left - 1).bit_length(),), dtype=torch.uint8).pin_memory()
                            with torch.cuda.stream(side):
                                tail_host[:left].copy_(text_dev[prev["b"]][at : at + left], non_blocking=True)
                            side.synchronize()
                            index_text_and_tokenize(tail_host.data_ptr(), prev["b"], at, left, left, True)
                        break
                elif batch is None:
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
        tok.synchronize()
        lap("close_and_drain")
    if stats and bool(torch.cat(stats).any()):
        load_dosage(vcf_file, chr_name, samples, ploidies, start, end, anc_allele_file, n_threads)
        raise ValueError(f"{vcf_file}: the GPU tokenizer flagged a line the host reader accepts")
    torch.cuda.current_stream(eng.device).wait_stream(side)
    torch.cuda.current_stream(eng.device).wait_stream(tok)
    pos = np.concatenate(pos_parts) if pos_parts else np.zeros(0, dtype=np.int32)
    dos = torch.cat(outs) if len(outs) > 1 else (outs[0] if outs else torch.empty((0, n), dtype=torch.int8, device=eng.device))
    lap("status_and_concat")
    if trace is not None:
        print(f"bgzf route ({index_from}), ms:", " ".join(f"{k}={1e3 * v:.1f}" for k, v in trace.items()), flush=True)
    return pos, dos, (int(n_match.value) if have_header else 0), (int(n_anc.value) if have_header else 0)


def scan_first_last_device(eng, vcf_file: str, chr_name: str):
    """First and last POS of the first contiguous run of ``chr_name`` in a bgzip file, found with the
    GPU-inflate pass (no sample column is tokenised): what ``ChunkGenerator`` needs before the windows
    can be laid out.  None when this route does not serve the file (not bgzip, lines it cannot index
    from their heads and would have to copy back whole, ...): the host scan does it then."""
    if os.environ.get("SAI_AMD_GPU_INFLATE", "1") == "0":
        return None
    cap = int(os.environ.get("SAI_AMD_INFLATE_BATCH", 0)) or _inflate_batch_for(vcf_file)
    try:
        got = _load_bgzf_device(eng, vcf_file, chr_name, [], [], None, None, None, None, cap, positions_only=True)
    except (_Fallback, _TextIndex):
        return None
    if got is None:
        return None
    pos = got[0]
    return (None, None) if pos.size == 0 else (int(pos[0]), int(pos[-1]))
