"""Shared single-window evaluation used by UStatistic / QStatistic."""

from __future__ import annotations

import numpy as np

from .. import _ffi
from .stat_utils import evaluate_sites, validate_thresholds


def run_single_window(stat, w, x, quantile, y_list, anc_allele_available):
    """Whole matrices = one window.  Returns (record, U site indices, Q site indices)."""
    import torch

    validate_thresholds(w, y_list, len(stat.src_gts_list))
    ploidy = [stat.ref_ploidy, stat.tgt_ploidy] + list(stat.src_ploidy_list)
    n_eff = min(len(stat.src_gts_list), max(len(ploidy) - 2, 0))
    prm = _ffi.make_params(w, x, quantile, y_list, anc_allele_available, n_src=n_eff)
    eng, tgt_freq, flags, _, _ = evaluate_sites(stat.ref_gts, stat.tgt_gts, stat.src_gts_list, ploidy, [prm])
    n_sites = int(flags.shape[1])
    lo = torch.zeros(1, dtype=torch.int32, device=eng.device)
    hi = torch.full((1,), n_sites, dtype=torch.int32, device=eng.device)
    res = eng.window_stats(tgt_freq, flags, [prm], lo, hi, pos=None, cap_hint=max(n_sites, 1))
    return res.records[0, 0], np.asarray(res.u_list(0, 0), dtype=np.int64), np.asarray(res.q_list(0, 0), dtype=np.int64)
