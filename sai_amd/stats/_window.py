"""Shared single-window evaluation used by UStatistic / QStatistic."""

from __future__ import annotations

import warnings

import numpy as np

from .. import _ffi
from ..engine import Engine
from .stat_utils import _check_ploidy, validate_thresholds


class PerWindowRouteWarning(UserWarning):
    """Many one-window calls in one process: the batched route is three orders of magnitude away."""


PER_WINDOW_WARN_AFTER = 64
_calls = 0


def note_per_window_call() -> None:
    """After ``PER_WINDOW_WARN_AFTER`` one-window statistic calls in this process, say ONCE what they cost.  The
    reference's per-window API (u_statistic.py:37-99, feature_preprocessor.py:63-191) is kept as it is, but here
    every call moves its window's matrices over PCIe: measured on MI355X 1 400-3 000 windows/s for C3-sized
    windows (profiles/r05_plugin_rate.txt) against 3.3 million windows/s of the resident route."""
    global _calls
    _calls += 1
    if _calls == PER_WINDOW_WARN_AFTER:
        warnings.warn(
            f"{PER_WINDOW_WARN_AFTER} one-window statistic calls (UStatistic / QStatistic.compute, FeaturePreprocessor.run) in "
            "this process: each uploads its window's genotype matrices over PCIe, about 1 400-3 000 windows/s for 2 000 x 2 000 "
            "windows on MI355X.  ChunkPreprocessor.run / FeaturePreprocessor.score_windows (what `sai score` uses) read a "
            "region once and score all of its windows in one resident pass: about 3 million windows/s, ~1 000 x this route.",
            PerWindowRouteWarning,
            stacklevel=4,
        )


def run_single_window(stat, w, x, quantile, y_list, anc_allele_available):
    """Whole matrices = one window.  Returns (record, U site indices, Q site indices): the
    matrices are uploaded (once per ``Engine.upload_scope``) and ``sai_single_window`` does the rest
    in one call -- fused site pass, window statistics over [0, n_sites), results on the host.

    ``x`` is None in a call that only wants Q, ``quantile`` is None in one that only wants U.  Inside an
    ``upload_scope`` that was told which thresholds go together (``hints``: FeaturePreprocessor.run knows all
    configured statistics of its window), U and Q of one parameter set are ONE device call: the first of the two
    computes both, the second takes its half from the scope."""
    note_per_window_call()
    validate_thresholds(w, y_list, len(stat.src_gts_list))
    ploidy = [stat.ref_ploidy, stat.tgt_ploidy] + list(stat.src_ploidy_list)
    for p in ploidy[: 2 + len(stat.src_gts_list)]:
        _check_ploidy(p)
    n_eff = min(len(stat.src_gts_list), max(len(ploidy) - 2, 0))  # zip() of stat_utils.py:116-119
    mats = [stat.ref_gts, stat.tgt_gts] + list(stat.src_gts_list[:n_eff])
    if len({int(np.shape(m)[0]) for m in mats}) != 1:
        raise ValueError("ref, tgt and src genotype matrices must have the same number of sites")
    eng = Engine.get()
    key = (tuple(id(m) for m in mats), tuple(int(p) for p in ploidy[: 2 + n_eff]), float(w),
           tuple((op, float(y)) for op, y in y_list), bool(anc_allele_available))  # fmt: skip
    scope = eng.window_scope()
    wanted = ("x", float(x)) if quantile is None else ("quantile", float(quantile))
    if scope is not None:
        have = scope["results"].get(key)
        if have is not None and have[0][wanted[0]] == wanted[1]:
            return have[1]
    both = {"x": 0.0 if x is None else float(x), "quantile": 0.5 if quantile is None else float(quantile)}
    hint = scope["hints"].get(key[1:]) if scope is not None else None
    if hint is not None and hint.get(wanted[0]) == wanted[1]:  # the window's other statistic of this parameter set
        both.update(hint)
    prm = _ffi.make_params(w, both["x"], both["quantile"], y_list, anc_allele_available, n_src=n_eff)
    pops = eng.tile_many(mats)
    if n_eff <= _ffi.SAI_FUSED_SRC:
        rec, idx_u, idx_q = eng.single_window(pops, ploidy[: 2 + n_eff], prm)
    else:  # more sources than a streaming pass takes (the reference loops over any number, stat_utils.py:114-119)
        rec, idx_u, idx_q = eng.single_window_unfused(pops, ploidy[: 2 + n_eff], prm)
    out = ({"n_sites": rec.n_sites, "u_count": rec.u_count, "n_cond": rec.n_cond, "n_cdd_q": rec.n_cdd_q, "q": rec.q}, idx_u, idx_q)
    if scope is not None:
        scope["results"][key] = (both, out)
    return out
