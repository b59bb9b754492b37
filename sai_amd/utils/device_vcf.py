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
