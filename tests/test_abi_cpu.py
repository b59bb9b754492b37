"""CPU-side checks of the C ABI: the library builds/loads, exports every symbol that
include/saihip.h declares, refuses to run without a device, and its host-side synthetic
generator equals the numpy definition.  No GPU compute is attempted here."""

import ctypes as C
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    g.build()
    from sai_amd import _ffi

    return _ffi.load()


def declared_functions():
    text = (ROOT / "include" / "saihip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sai_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    from sai_amd import _ffi

    names = declared_functions()
    assert len(names) >= 15
    assert sorted(_ffi.SIGNATURES) == names
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in saihip.h but not exported"
    assert lib.sai_abi_version() == _ffi.SAI_ABI_VERSION
    assert lib.sai_build_arch() == b"gfx950"
    assert C.sizeof(_ffi.SaiWindowRecord) == 24
    assert C.sizeof(_ffi.SaiParams) == 32 + _ffi.SAI_MAX_SRC * 20 == 312  # op (4) + y (8) + one_minus_y (8) per source
    assert C.sizeof(_ffi.SaiPop) == 16


def test_tiled_bytes(lib):
    assert lib.sai_tiled_bytes(0, 5) == 0
    assert lib.sai_tiled_bytes(1, 5) == 5 * 64
    assert lib.sai_tiled_bytes(64, 1000) == 64000
    assert lib.sai_tiled_bytes(65, 1000) == 128000
    assert lib.sai_tiled_bytes(-1, 3) == -1


def test_fails_loudly_without_gpu(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from sai_amd import _ffi
    from sai_amd.engine import Engine

    ctx = C.c_void_p()
    rc = lib.sai_ctx_create(0, C.byref(ctx))
    assert rc != 0 and not ctx.value
    assert b"no CPU fallback" in lib.sai_last_error() or lib.sai_last_error()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(0)
    from sai_amd.stats import UStatistic

    g = np.zeros((2, 2), dtype=np.int64)
    with pytest.raises(RuntimeError):
        UStatistic(ref_gts=g, tgt_gts=g, src_gts_list=[g], ref_ploidy=2, tgt_ploidy=2, src_ploidy_list=[2]).compute(
            pos=np.array([1, 2]), w=0.5, x=0.5, y_list=[("=", 1.0)], anc_allele_available=True
        )
    assert _ffi.SaiHipError(-3, "x").status == -3


def test_plan_and_plane_entry_points_refuse_bad_arguments(lib):
    """The round-3 entry points without a GPU: sizes, NULL handles and a ctx that cannot exist are
    answered with a status and a message, never a crash."""
    from sai_amd import _ffi

    assert lib.sai_plane_words(64, 1) == 3 and lib.sai_plane_words(65, 2) == 12 and lib.sai_plane_words(0, 5) == 0
    assert lib.sai_plane_words(-1, 1) == -1 and lib.sai_plane_words(10, -1) == -1
    plan = C.c_void_p()
    assert lib.sai_plan_create(None, C.byref(plan)) == _ffi.SAI_ERR_ARG and not plan.value
    assert lib.sai_plan_run(None, None) == _ffi.SAI_ERR_ARG and b"NULL" in lib.sai_last_error()
    assert lib.sai_plan_add_copy_to_host(None, None, None, 0) == _ffi.SAI_ERR_ARG
    assert lib.sai_plan_add_copy_to_host(None, None, None, -5) == _ffi.SAI_ERR_ARG
    assert lib.sai_plan_add_window_bounds(None, None, 0, 0, None, None, None, None, None, None) == _ffi.SAI_ERR_ARG
    assert lib.sai_plan_destroy(None) == 0
    a = C.c_int64()
    assert lib.sai_bgzf_stream_region(None, C.byref(a), C.byref(a), C.byref(a)) == _ffi.SAI_ERR_ARG


def test_params_packing():
    from sai_amd import _ffi

    p = _ffi.make_params(0.01, 0.5, 0.95, [("=", 0.9), (">=", 0.2)], False)
    assert (p.w, p.x, p.quantile, p.n_src, p.anc_allele_available) == (0.01, 0.5, 0.95, 2, 0)
    assert list(p.op)[:2] == [0, 4]
    assert p.one_minus_y[0] == 1 - 0.9 and p.one_minus_y[0] != 0.1  # the f64 mirror, not the decimal one
    seven = _ffi.make_params(0.1, 0.1, 0.5, [("=", 1.0)] * 7, True)  # more than a streaming pass takes: counts in groups + site_flags
    assert seven.n_src == 7 > _ffi.SAI_FUSED_SRC and list(seven.y)[:8] == [1.0] * 7 + [0.0]
    with pytest.raises(ValueError):
        _ffi.make_params(0.1, 0.1, 0.5, [("=", 1.0)] * (_ffi.SAI_MAX_SRC + 1), True)


@pytest.mark.parametrize(
    "pop_stream,n_ind,ploidy,mpm", [(0, 33, 2, 0), (1, 17, 2, 1000), (2, 2, 2, 20000), (3, 3, 4, 0), (1, 8, 1, 500000), (0, 5, 3, 0)]
)
def test_host_synth_equals_numpy_definition(lib, pop_stream, n_ind, ploidy, mpm):
    from oracle import synth_numpy as S
    from sai_amd import _ffi

    seed, chrom, site0, n_sites = 20260630 + pop_stream, 7, 99000, 3000
    got = np.empty((n_sites, n_ind), dtype=np.int8)
    _ffi.check(lib.sai_synth_fill_host(seed, chrom, site0, n_sites, pop_stream, n_ind, ploidy, mpm,
                                       got.ctypes.data_as(C.c_void_p)))  # fmt: skip
    exp = S.genotypes(seed, chrom, site0, n_sites, pop_stream, n_ind, ploidy, mpm)
    assert np.array_equal(got, exp)
    gaps = np.empty(n_sites, dtype=np.int32)
    _ffi.check(lib.sai_synth_gaps_host(seed, chrom, site0, n_sites, gaps.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(gaps, S.gaps(seed, chrom, site0, n_sites))


def test_synth_statistics_are_sane(lib):
    """The synthetic chromosome must exercise the statistics: ~0.1 % introgressed sites with
    ref fixed 0 / src fixed ALT, rare-variant background, and U candidates at default thresholds."""
    from oracle import synth_numpy as S

    seed, n = 20260632, 200000
    ref = S.genotypes(seed, 1, 0, n, 0, 20, 2)
    tgt = S.genotypes(seed, 1, 0, n, 1, 20, 2)
    src = S.genotypes(seed, 1, 0, n, 2, 2, 2)
    intro = (ref.sum(1) == 0) & (src.min(1) == 2) & (tgt.mean(1) / 2 > 0.1)
    assert 100 < intro.sum() < 400
    assert 0.15 < (ref.mean() / 2) < 0.25  # E[u^4] = 0.2
    g = S.gaps(seed, 1, 0, n)
    assert 24 < g.mean() < 26
