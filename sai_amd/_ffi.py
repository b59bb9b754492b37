"""ctypes binding of libsaihip.so (include/saihip.h).

There is no CPU fallback: if the shared library is missing or no gfx950 device
is usable, the calls raise.  ``load()`` only needs the library file (and the HIP
runtime it links against), so symbol checks work on a machine without a GPU.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_NAME = "libsaihip.so"
LIB_PATH = Path(__file__).resolve().parent / "lib" / LIB_NAME

SAI_TILE_SITES = 64
SAI_MAX_SRC = 14  # source populations of a parameter set
SAI_FUSED_SRC = 6  # ... of which a streaming pass takes this many per call (more: counts in groups + site_flags)
SAI_MAX_SETS = 20
SAI_FUSED_SETS = SAI_MAX_SETS
SAI_PLANES_PER_SET = 3
SAI_DD_FUSED_ROWS = 4  # source individuals whose DD terms can ride along the site pass
SAI_ERR_ARG = -1
SAI_ERR_HIP = -2
SAI_ERR_NO_DEVICE = -3
SAI_ERR_UNSUPPORTED = -4  # enum sai_status
FREQ_MODES = {"dense": 0, "candidates": 1}  # enum sai_freq_mode
SAI_ABI_VERSION = 16

OPS = {"=": 0, "<": 1, ">": 2, "<=": 3, ">=": 4}


class SaiHipError(RuntimeError):
    """A libsaihip call returned a non-zero status."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libsaihip error {status}: {message}")
        self.status = status


class SaiPop(C.Structure):
    _fields_ = [("tiles", C.c_void_p), ("n_ind", C.c_int32), ("ploidy", C.c_int32)]


class SaiParams(C.Structure):
    _fields_ = [
        ("w", C.c_double),
        ("x", C.c_double),
        ("quantile", C.c_double),
        ("n_src", C.c_int32),
        ("anc_allele_available", C.c_int32),
        ("op", C.c_int32 * SAI_MAX_SRC),
        ("y", C.c_double * SAI_MAX_SRC),
        ("one_minus_y", C.c_double * SAI_MAX_SRC),
    ]


class SaiDdRows(C.Structure):
    _fields_ = [("first_pop", C.c_int32), ("n_pops", C.c_int32), ("absdiff", C.c_void_p)]


class SaiTextColumn(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride_bytes", C.c_int64), ("kind", C.c_int32), ("reserved", C.c_int32)]


class SaiLogRows(C.Structure):
    _fields_ = [("counts_host", C.c_void_p), ("count_stride_bytes", C.c_int64), ("offsets_host", C.c_void_p),
                ("offset_stride_words", C.c_int64), ("positions_host", C.c_void_p), ("position_bytes", C.c_int32),
                ("fd", C.c_int32)]  # fmt: skip


SAI_MAX_LOGS = 8


class SaiWindowRecord(C.Structure):
    _fields_ = [
        ("n_sites", C.c_int32),
        ("u_count", C.c_int32),
        ("n_cond", C.c_int32),
        ("n_cdd_q", C.c_int32),
        ("q", C.c_double),
    ]


_p = C.c_void_p
_i32, _i64, _u64 = C.c_int32, C.c_int64, C.c_uint64

# name -> (restype, argtypes); the single source of truth for the symbol check in tests
SIGNATURES = {
    "sai_abi_version": (C.c_int, []),
    "sai_build_arch": (C.c_char_p, []),
    "sai_last_error": (C.c_char_p, []),
    "sai_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sai_device_identity": (C.c_int, [C.c_int, C.c_char_p, _i32, C.c_char_p, _i32]),
    "sai_ctx_create": (C.c_int, [C.c_int, C.POINTER(_p)]),
    "sai_ctx_destroy": (C.c_int, [_p]),
    "sai_tiled_bytes": (_i64, [_i64, _i32]),
    "sai_tile_from_site_major": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _p]),
    "sai_site_counts": (C.c_int, [_p, _i64, _i32, C.POINTER(SaiPop), _p, _p]),
    "sai_plane_words": (_i64, [_i64, _i32]),
    "sai_site_pass": (
        C.c_int,
        [_p, _i64, _i32, C.POINTER(SaiPop), _p, _i32, C.POINTER(SaiParams), _i32, _p, _p, _i64, _p],
    ),
    "sai_site_pass_dd": (
        C.c_int,
        [_p, _i64, _i32, C.POINTER(SaiPop), _p, _i32, C.POINTER(SaiParams), _i32, _p, _p, _i64, C.POINTER(SaiDdRows), _p],
    ),
    "sai_plan_add_site_pass_dd": (
        C.c_int,
        [_p, _i64, _i32, C.POINTER(SaiPop), _p, _i32, C.POINTER(SaiParams), _i32, _p, _p, _i64, C.POINTER(SaiDdRows)],
    ),
    "sai_site_flags": (
        C.c_int,
        [_p, _i64, _i32, C.POINTER(_i32), _p, _i32, C.POINTER(SaiParams), _p, _p, _i64, _p, _p],
    ),
    "sai_window_bounds": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _p]),
    "sai_window_bounds_seg": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "sai_window_total_words": (_i64, [_i32, _i32]),
    "sai_window_stats": (
        C.c_int,
        [_p, _i64, _p, _p, _i64, _i32, C.POINTER(SaiParams), _i32, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p],
    ),
    "sai_plan_create": (C.c_int, [_p, C.POINTER(_p)]),
    "sai_plan_destroy": (C.c_int, [_p]),
    "sai_plan_run": (C.c_int, [_p, _p]),
    "sai_plan_set_pass_events": (C.c_int, [_p, _p, _p]),
    "sai_event_create": (C.c_int, [_p, C.POINTER(_p)]),
    "sai_event_destroy": (C.c_int, [_p]),
    "sai_event_synchronize": (C.c_int, [_p]),
    "sai_event_query": (C.c_int, [_p, C.POINTER(_i32)]),
    "sai_event_elapsed_ms": (C.c_int, [_p, _p, C.POINTER(C.c_float)]),
    "sai_plan_add_site_counts": (C.c_int, [_p, _i64, _i32, C.POINTER(SaiPop), _p]),
    "sai_plan_add_site_pass": (
        C.c_int,
        [_p, _i64, _i32, C.POINTER(SaiPop), _p, _i32, C.POINTER(SaiParams), _i32, _p, _p, _i64, _i32],
    ),
    "sai_plan_add_site_flags": (C.c_int, [_p, _i64, _i32, C.POINTER(_i32), _p, _i32, C.POINTER(SaiParams), _p, _p, _i64]),
    "sai_plan_add_window_bounds": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "sai_plan_add_window_stats": (
        C.c_int,
        [_p, _i64, _p, _p, _i64, _i32, C.POINTER(SaiParams), _i32, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _p],
    ),
    "sai_plan_add_copy_to_host": (C.c_int, [_p, _p, _p, _i64]),
    "sai_single_window": (
        C.c_int,
        [_p, _i64, _i32, C.POINTER(SaiPop), C.POINTER(SaiParams), C.POINTER(SaiWindowRecord), _p, _p, _p],
    ),
    "sai_site_freqs": (C.c_int, [_p, _i64, _i32, C.POINTER(_i32), _p, _p, _p]),
    "sai_window_fourpop": (C.c_int, [_p, _i64, _i32, _i32, _p, _i32, _p, _p, _p, _p, _p]),
    "sai_pattern_sum": (C.c_int, [_p, _i64, _p, _p, _p, _p, _i32, _p, _p]),
    "sai_site_absdiff": (C.c_int, [_p, _i64, C.POINTER(SaiPop), C.POINTER(SaiPop), _p, _p]),
    "sai_window_dd": (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _i32, _i32, _p, _p, _p, _p, _p]),
    "sai_packed2_bytes": (_i64, [_i64, _i32]),
    "sai_pack2_from_tiles": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p]),
    "sai_site_pass_packed2": (
        C.c_int,
        [_p, _i64, _i32, C.POINTER(SaiPop), _p, _i32, C.POINTER(SaiParams), _i32, _p, _p, _i64, _p],
    ),
    "sai_synth_fill": (C.c_int, [_p, _u64, _i32, _i64, _i64, _i32, _i32, _i32, _i32, _p, _p]),
    "sai_synth_fill_host": (C.c_int, [_u64, _i32, _i64, _i64, _i32, _i32, _i32, _i32, _p]),
    "sai_synth_gaps_host": (C.c_int, [_u64, _i32, _i64, _i64, _p]),
    "sai_synth_gaps": (C.c_int, [_p, _u64, _i32, _i64, _i64, _p, _p]),
    "sai_narrow_to_int8": (C.c_int, [_p, _i32, _i32, _i64, _i64, _i64, _p, _i32]),
    "sai_probe_stream_read": (C.c_int, [_p, _p, _i64, _p, _p]),
    "sai_vcf_scan": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_i64), C.POINTER(_i64)]),
    "sai_vcf_load": (
        C.c_int,
        [C.c_char_p, C.c_char_p, _i64, _i64, _i32, C.POINTER(C.c_char_p), C.POINTER(_i32), C.c_char_p, _i32, C.POINTER(_p)],
    ),
    "sai_vcf_stream_open": (
        C.c_int,
        [C.c_char_p, C.c_char_p, _i64, _i64, _i32, C.POINTER(C.c_char_p), C.POINTER(_i32), C.c_char_p, _i32, _p, _p, _i64,
         C.POINTER(_p)],
    ),
    "sai_vcf_stream_next": (
        C.c_int,
        [_p, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p),
         C.POINTER(_p), C.POINTER(_i32)],
    ),
    "sai_vcf_stream_selection": (C.c_int, [_p, _p, _i32, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64)]),
    "sai_vcf_stream_close": (C.c_int, [_p]),
    "sai_inflate_bgzf": (C.c_int, [_p, _p, _i64, _p, _i32, _p, _i64, _p, _p]),
    "sai_text_line_starts": (C.c_int, [_p, _p, _i64, _i64, _p, _p, _p, _p, _p]),
    "sai_text_line_heads": (C.c_int, [_p, _p, _i64, _p, _i64, _i32, _p, _p]),
    "sai_bgzf_stream_open": (
        C.c_int,
        [C.c_char_p, C.c_char_p, _i64, _i64, _i32, C.POINTER(C.c_char_p), C.POINTER(_i32), C.c_char_p, _i32, _p, _p, _i64,
         _i64, C.POINTER(_p)],
    ),
    "sai_bgzf_stream_next": (
        C.c_int, [_p, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_p), C.POINTER(_i64), C.POINTER(_i32)],
    ),
    "sai_vcf_index_text": (
        C.c_int,
        [_p, _p, _i64, _i64, _p, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p),
         C.POINTER(_p), C.POINTER(_p), C.POINTER(_i32)],
    ),
    "sai_bgzf_stream_release": (C.c_int, [_p]),
    "sai_vcf_index_heads": (
        C.c_int,
        [_p, _p, _i32, _p, _p, _i64, C.POINTER(_i64), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p),
         C.POINTER(_p), C.POINTER(_i32)],
    ),
    "sai_bgzf_stream_region": (C.c_int, [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "sai_bgzf_stream_selection": (C.c_int, [_p, _p, _i32, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64)]),
    "sai_bgzf_stream_close": (C.c_int, [_p]),
    "sai_tokenize_gt": (C.c_int, [_p, _p, _i64, _i64, _p, _p, _p, _p, _i32, _p, _i32, _p, _p, _p, _p]),
    "sai_format_score_rows": (
        C.c_int,
        [C.c_char_p, C.c_char_p, _i32, _p, _p, _i32, C.POINTER(SaiTextColumn), C.POINTER(_p)],
    ),
    "sai_format_log_rows": (C.c_int, [C.c_char_p, _i32, _p, _p, _i64, _p, _i64, _p, _i32, C.POINTER(_p)]),
    "sai_format_doubles": (C.c_int, [_p, _i64, C.POINTER(_p)]),
    "sai_write_window_rows": (
        C.c_int,
        [C.c_char_p, C.c_char_p, _i32, _p, _p, _i32, C.POINTER(SaiTextColumn), _i32, _i32, C.POINTER(SaiLogRows), C.POINTER(_i64)],
    ),
    "sai_text_data": (_p, [_p, C.POINTER(_i64)]),
    "sai_text_free": (C.c_int, [_p]),
    "sai_vcf_block_info": (C.c_int, [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "sai_vcf_block_copy": (C.c_int, [_p, _p, _p]),
    "sai_vcf_block_free": (C.c_int, [_p]),
}

# entry points that never touch the GPU (host_core.cpp, vcf_ingest.cpp)
HOST_SYMBOLS = (
    "sai_abi_version", "sai_build_arch", "sai_last_error", "sai_synth_fill_host", "sai_synth_gaps_host",
    "sai_narrow_to_int8", "sai_vcf_scan", "sai_vcf_load", "sai_vcf_block_info", "sai_vcf_block_copy", "sai_vcf_block_free",
    "sai_vcf_stream_open", "sai_vcf_stream_next", "sai_vcf_stream_selection", "sai_vcf_stream_close",
    "sai_bgzf_stream_open", "sai_bgzf_stream_next", "sai_bgzf_stream_release", "sai_bgzf_stream_region", "sai_vcf_index_text", "sai_vcf_index_heads", "sai_bgzf_stream_selection", "sai_bgzf_stream_close",
    "sai_format_score_rows", "sai_format_log_rows", "sai_format_doubles", "sai_text_data", "sai_text_free",
    "sai_write_window_rows",
)  # fmt: skip

_lib = None
_host_lib = None


def load() -> C.CDLL:
    """Load libsaihip.so (once) and declare every prototype.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("SAI_AMD_LIB", LIB_PATH))
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root). "
            "sai_amd has no CPU fallback."
        )
    # torch ships its own libamdhip64.so (soname libamdhip64.so.7).  It must be in the process
    # BEFORE libsaihip is loaded so that libsaihip's NEEDED libamdhip64.so.7 resolves to that
    # already-loaded runtime: one HIP runtime per process, shared with the torch tensors and
    # streams handed to the C ABI.  Loaded the other way round, torch would bring in a second
    # runtime next to the system one and device pointers would cross runtimes.
    import torch  # noqa: F401

    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.sai_abi_version() != SAI_ABI_VERSION:
        raise RuntimeError(f"{path}: ABI {lib.sai_abi_version()} != expected {SAI_ABI_VERSION}")
    _lib = lib
    return lib


def load_host() -> C.CDLL:
    """The library that serves the host-only entry points (``HOST_SYMBOLS``): libsaihip.so itself,
    or -- when ``SAI_AMD_HOST_LIB`` names one -- a separate build of host_core.cpp + vcf_ingest.cpp
    (the sanitizer build of the CPU test suite), loaded without the HIP runtime or torch."""
    global _host_lib
    if _host_lib is not None:
        return _host_lib
    path = os.environ.get("SAI_AMD_HOST_LIB")
    if not path:
        _host_lib = load()
        return _host_lib
    lib = C.CDLL(path)
    for name in HOST_SYMBOLS:
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = SIGNATURES[name]
    if lib.sai_abi_version() != SAI_ABI_VERSION:
        raise RuntimeError(f"{path}: ABI {lib.sai_abi_version()} != expected {SAI_ABI_VERSION}")
    _host_lib = lib
    return lib


def check(status: int, lib=None) -> None:
    if status != 0:
        raise SaiHipError(status, (lib or load()).sai_last_error().decode("utf-8", "replace"))


def make_params(w, x, quantile, y_list, anc_allele_available, n_src=None) -> SaiParams:
    """Pack one parameter set.  ``y_list`` = [(op, y), ...]; ``1 - y`` is evaluated here in
    Python f64, exactly the value the reference compares with (stat_utils.py:150)."""
    y_list = list(y_list)
    if n_src is None:
        n_src = len(y_list)
    if n_src > SAI_MAX_SRC:
        raise ValueError(f"at most {SAI_MAX_SRC} source populations are supported")
    p = SaiParams()
    p.w, p.x, p.quantile = float(w), float(x), float(quantile)
    p.n_src = n_src
    p.anc_allele_available = 1 if anc_allele_available else 0
    for k, (op, y) in enumerate(y_list[:n_src]):
        p.op[k] = OPS[op]
        p.y[k] = float(y)
        p.one_minus_y[k] = float(1 - y)
    return p
