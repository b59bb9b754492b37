"""ctypes front end of the native VCF tokenizer in libsaihip (sai_vcf_scan / sai_vcf_load,
sai_amd/csrc/vcf_ingest.cpp).  Host-side, except that the scan of a bgzip file is handed to the
GPU-inflate pass when a GPU is present."""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from .. import _ffi


def default_threads() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def scan_first_last(vcf_file: str, chr_name: str) -> tuple[Optional[int], Optional[int]]:
    """First and last POS of the first contiguous run of ``chr_name`` (None, None if absent).  With a
    usable tabix index next to the file the host scan reads two records (the chromosome's first chunk
    and its last one) whatever the size of the file -- every rank of a sharded run asks this question.
    Otherwise a bgzip file is scanned with the GPU-inflate pass when a GPU is there (the host would
    inflate the whole file for it: 53 ms against 12 for 480 MB of text); everything else, and every
    machine without a GPU, takes the host scan."""
    if str(vcf_file).endswith((".gz", ".bgz")) and os.environ.get("SAI_AMD_INGEST") != "host" and not _index_usable(vcf_file):
        got = _scan_on_device(vcf_file, chr_name)
        if got is not None:
            return got
    lib = _ffi.load_host()
    first, last = C.c_int64(-1), C.c_int64(-1)
    _check_io(lib.sai_vcf_scan(os.fsencode(vcf_file), str(chr_name).encode(), C.byref(first), C.byref(last)))
    if first.value < 0:
        return None, None
    return int(first.value), int(last.value)


def _index_usable(vcf_file: str) -> bool:
    """``<vcf>.tbi`` exists and is not older than its file (whole seconds, the rule of load_tbi in
    ingest_base.hpp and of htslib)."""
    try:
        return int(os.stat(str(vcf_file) + ".tbi").st_mtime) >= int(os.stat(vcf_file).st_mtime)
    except OSError:
        return False


def _scan_on_device(vcf_file: str, chr_name: str):
    try:
        import torch

        if not torch.cuda.is_available():
            return None
        from ..engine import Engine
        from .device_vcf import scan_first_last_device

        return scan_first_last_device(Engine.get(), str(vcf_file), chr_name)
    except (ImportError, ValueError):  # a file the indexer refuses: the host scan (and later the load) says why
        return None


def _check_io(status: int) -> None:
    """I/O and format problems of the ingest surface as ValueError, like the reference's readers."""
    if status != 0:
        raise ValueError(_ffi.load_host().sai_last_error().decode("utf-8", "replace"))


def load_dosage(vcf_file: str, chr_name: str, samples: Sequence[str], ploidies: Sequence[int],
                start: Optional[int] = None, end: Optional[int] = None, anc_allele_file: Optional[str] = None,
                n_threads: Optional[int] = None) -> tuple[np.ndarray, np.ndarray, int, int]:  # fmt: skip
    """(pos int32 [n], dosage int8 [n][len(samples)], n_matched, n_anc_entries) for one region;
    with ``anc_allele_file`` the rows are already polarised (kept / flipped)."""
    lib = _ffi.load_host()
    n = len(samples)
    names = (C.c_char_p * n)(*[s.encode() for s in samples])
    pl = (C.c_int32 * n)(*[int(p) for p in ploidies])
    blk = C.c_void_p()
    _check_io(
        lib.sai_vcf_load(
            os.fsencode(vcf_file), str(chr_name).encode(), -1 if start is None else int(start),
            -1 if end is None else int(end), n, names, pl,
            os.fsencode(anc_allele_file) if anc_allele_file else None, n_threads or default_threads(), C.byref(blk),
        )
    )  # fmt: skip
    try:
        n_rec, n_match, n_anc = C.c_int64(), C.c_int64(), C.c_int64()
        _check_io(lib.sai_vcf_block_info(blk, C.byref(n_rec), C.byref(n_match), C.byref(n_anc)))
        pos = np.empty(n_rec.value, dtype=np.int32)
        dos = np.empty((n_rec.value, n), dtype=np.int8)
        _check_io(lib.sai_vcf_block_copy(blk, pos.ctypes.data_as(C.c_void_p), dos.ctypes.data_as(C.c_void_p)))
    finally:
        lib.sai_vcf_block_free(blk)
    return pos, dos, int(n_match.value), int(n_anc.value)
