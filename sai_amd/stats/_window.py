"""Shared single-window evaluation used by UStatistic / QStatistic."""

from __future__ import annotations

import numpy as np

from .. import _ffi
from ..engine import Engine
from .stat_utils import _check_ploidy, validate_thresholds


def run_single_window(stat, w, x, quantile, y_list, anc_allele_available):
    """Whole matrices = one window.  Returns (record, U site indices, Q site indices): the
    matrices are uploaded (once per ``Engine.upload_scope``) and ``sai_single_window`` does the rest
    in one call -- fused site pass, window statistics over [0, n_sites), results on the host."""
    validate_thresholds(w, y_list, len(stat.src_gts_list))
    ploidy = [stat.ref_ploidy, stat.tgt_ploidy] + list(stat.src_ploidy_list)
    for p in ploidy[: 2 + len(stat.src_gts_list)]:
        _check_ploidy(p)
    n_eff = min(len(stat.src_gts_list), max(len(ploidy) - 2, 0))  # zip() of stat_utils.py:116-119
    mats = [stat.ref_gts, stat.tgt_gts] + list(stat.src_gts_list[:n_eff])
    if len({int(np.shape(m)[0]) for m in mats}) != 1:
        raise ValueError("ref, tgt and src genotype matrices must have the same number of sites")
    prm = _ffi.make_params(w, x, quantile, y_list, anc_allele_available, n_src=n_eff)
    eng = Engine.get()
    pops = eng.tile_many(mats)
    rec, idx_u, idx_q = eng.single_window(pops, ploidy[: 2 + n_eff], prm)
    return {"n_sites": rec.n_sites, "u_count": rec.u_count, "n_cond": rec.n_cond, "n_cdd_q": rec.n_cdd_q, "q": rec.q}, idx_u, idx_q
