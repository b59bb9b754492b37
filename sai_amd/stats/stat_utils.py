"""calc_freq / compute_matching_loci with sai's signatures, evaluated on the GPU.

Mirrors sai/stats/stat_utils.py:26-168: same arguments, same return values, same exceptions
and messages; the arithmetic runs in libsaihip (site_counts + site_flags kernels).
"""

from __future__ import annotations

from typing import Sequence

import numpy as np

from .. import _ffi
from ..engine import FLAG_COND, Engine

_OPS = ("=", "<", ">", "<=", ">=")


def _check_ploidy(ploidy) -> None:
    # stat_utils.py:42-43
    if not isinstance(ploidy, int) or ploidy <= 0:
        raise ValueError("ploidy must be a positive integer.")


def validate_thresholds(w, y_list, n_src_gts) -> None:
    """stat_utils.py:99-111, same order and messages."""
    if not (0 <= w <= 1):
        raise ValueError("Parameters w must be within the range [0, 1].")
    for op, y in y_list:
        if not (0 <= y <= 1):
            raise ValueError(f"Invalid value in y_list: {y}. within the range [0, 1].")
        if op not in _OPS:
            raise ValueError(
                f"Invalid operator in y_list: {op}. Must be '=', '<', '>', '<=', or '>='."
            )
    if n_src_gts != len(y_list):
        raise ValueError("The length of src_gts_list and y_list must match.")


def calc_freq(gts: np.ndarray, ploidy: int = 1) -> np.ndarray:
    """Frequency of allele 1 per site, NaN where no individual is called
    (stat_utils.py:26-52)."""
    _check_ploidy(ploidy)
    eng = Engine.get()
    pop = eng.tile(gts)
    counts = eng.site_counts([pop, pop])
    dummy = _ffi.make_params(1.0, 0.0, 0.5, [], True, n_src=0)
    tgt_freq, _, _ = eng.site_flags(counts, [ploidy, ploidy], [dummy])
    return tgt_freq.cpu().numpy()


def evaluate_sites(ref_gts, tgt_gts, src_gts_list, ploidy: Sequence[int], sets, want_adj=False):
    """Upload one window and run site_counts + site_flags for the given parameter sets.
    The sources are paired with ``ploidy[2:]`` by zip, like stat_utils.py:116-119."""
    for p in ploidy[: 2 + len(src_gts_list)]:
        _check_ploidy(p)
    n_eff = min(len(src_gts_list), max(len(ploidy) - 2, 0))
    mats = [ref_gts, tgt_gts] + list(src_gts_list[:n_eff])
    n_sites = {int(np.shape(m)[0]) for m in mats}
    if len(n_sites) != 1:
        raise ValueError("ref, tgt and src genotype matrices must have the same number of sites")
    eng = Engine.get()
    pops = [eng.tile(m) for m in mats]
    counts = eng.site_counts(pops)
    tgt_freq, flags, adj = eng.site_flags(counts, list(ploidy[: 2 + n_eff]), sets, want_adj=want_adj)
    return eng, tgt_freq, flags, adj, n_eff


def compute_matching_loci(
    ref_gts: np.ndarray,
    tgt_gts: np.ndarray,
    src_gts_list: list[np.ndarray],
    w: float,
    y_list: list[tuple[str, float]],
    ploidy: list[int],
    anc_allele_available: bool,
) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(ref_freq, tgt_freq, condition) with polarity inversion applied (stat_utils.py:55-168)."""
    validate_thresholds(w, y_list, len(src_gts_list))
    n_eff = min(len(src_gts_list), max(len(ploidy) - 2, 0))
    prm = _ffi.make_params(w, 0.0, 0.5, y_list, anc_allele_available, n_src=n_eff)
    eng, _, planes, adj, _ = evaluate_sites(ref_gts, tgt_gts, src_gts_list, ploidy, [prm], want_adj=True)
    adj = adj.cpu().numpy()
    cond = (eng.flag_bytes(planes, adj.shape[2], [prm])[0].cpu().numpy() & FLAG_COND).astype(bool)
    return adj[0, 0].copy(), adj[0, 1].copy(), cond


def calc_four_pops_freq(ref_gts, tgt_gts, src_gts, out_gts=None, ref_ploidy: int = 1, tgt_ploidy: int = 1,
                        src_ploidy: int = 1, out_ploidy: int = 1):  # fmt: skip
    """(ref_freq, tgt_freq, src_freq, out_freq) of one window (stat_utils.py:171-217): ``calc_freq``
    of each population -- one site_counts + one site_freqs launch for all of them -- and an outgroup
    frequency of 0 everywhere when ``out_gts`` is None."""
    mats, ploidy = [ref_gts, tgt_gts, src_gts], [ref_ploidy, tgt_ploidy, src_ploidy]
    if out_gts is not None:
        mats.append(out_gts)
        ploidy.append(out_ploidy)
    for p in ploidy:
        _check_ploidy(p)
    eng = Engine.get()
    pops = [eng.tile(m) for m in mats]
    if len({p.n_sites for p in pops}) != 1:
        raise ValueError("genotype matrices must have the same number of sites")
    freqs = eng.site_freqs(eng.site_counts(pops), ploidy).cpu().numpy()
    out_freq = freqs[3].copy() if out_gts is not None else np.zeros_like(freqs[0])
    return freqs[0].copy(), freqs[1].copy(), freqs[2].copy(), out_freq


def calc_pattern_sum(ref_freq, tgt_freq, src_freq, out_freq, pattern: str) -> float:
    """Sum over sites of the product picked by a four-letter pattern over (ref, tgt, src, out):
    'b' takes the frequency, 'a' one minus it (stat_utils.py:220-272, same ValueErrors); the
    products and the sum are formed on the GPU in numpy's order, so the result is np.sum's double."""
    if len(pattern) != 4:
        raise ValueError("Pattern must be a four-character string.")
    bits = 0
    for k, c in enumerate(pattern.lower()):
        if c == "b":
            bits |= 1 << k
        elif c != "a":
            raise ValueError(f"Invalid character '{c}' in pattern. Only 'a' and 'b' allowed.")
    return Engine.get().pattern_sum(ref_freq, tgt_freq, src_freq, out_freq, bits)
