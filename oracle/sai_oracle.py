"""numpy restatement of sai's sliding-window U/Q path -- TEST INFRASTRUCTURE ONLY.

This file is the parity oracle and the timed CPU baseline ("port") for the HIP
path in ``sai_amd``.  It restates, in plain numpy and with the reference's own
structure (per-window ``[sites][individuals]`` integer matrices, frequencies
recomputed by every statistic, full-chromosome position masks per window), the
algorithm of xin-huang/sai 1.1.2.  Every function cites the reference lines it
follows (paths relative to the reference checkout).

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against ``tests/golden/*.json``: the known answers of the reference's own
tests (tests/stats/test_u_statistic.py, test_q_statistic.py,
test_stat_utils.py, tests/utils/test_utils.py:423-447,
tests/generators/test_chunk_generator.py:38-40,
tests/preprocessors/test_feature_preprocessor.py:158,223, tests/test_sai.py:63)
and outputs of the reference itself run on seeded inputs by
``tests/golden/make_golden.py``.

Product code must never import this module.
"""

from __future__ import annotations

import itertools
import operator
from typing import Any, Iterator, Optional, Sequence

import numpy as np

# ---------------------------------------------------------------------------
# a1  calc_freq                                   sai/stats/stat_utils.py:26-52
# ---------------------------------------------------------------------------


def allele_freq(gts: np.ndarray, ploidy: int = 1) -> np.ndarray:
    """Per-site frequency of allele 1 with missing calls (negative) ignored.

    sai/stats/stat_utils.py:42-43 (ploidy check), :45-46 (missing mask and
    called count), :48-49 (dosage sum as f64, denominator = called * ploidy),
    :51-52 (NaN where nothing is called).
    """
    if not isinstance(ploidy, int) or ploidy <= 0:
        raise ValueError("ploidy must be a positive integer.")
    g = np.asarray(gts)
    present = g >= 0
    n_called = present.sum(axis=1)
    dosage = np.where(present, g, 0).sum(axis=1, dtype=np.float64)
    denom = n_called * ploidy
    freq = np.full(g.shape[0], np.nan, dtype=np.float64)
    np.divide(dosage, denom, out=freq, where=denom > 0)
    return freq


# ---------------------------------------------------------------------------
# a2  compute_matching_loci                     sai/stats/stat_utils.py:55-168
# ---------------------------------------------------------------------------

_COMPARE = {
    "=": operator.eq,
    "<": operator.lt,
    ">": operator.gt,
    "<=": operator.le,
    ">=": operator.ge,
}


def _unit_interval(f: np.ndarray) -> np.ndarray:
    return np.isfinite(f) & (f >= 0) & (f <= 1)


def matching_loci(
    ref_gts: np.ndarray,
    tgt_gts: np.ndarray,
    src_gts_list: Sequence[np.ndarray],
    w: float,
    y_list: Sequence[tuple[str, float]],
    ploidy: Sequence[int],
    anc_allele_available: bool,
) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(ref_freq, tgt_freq, condition) with polarity inversion applied.

    Validation and messages: stat_utils.py:99-111.  Frequencies :114-119 (the
    sources are zipped with ``ploidy[2:]``, so the shorter list wins).  ``valid``
    :121-130.  Source matching against ``y`` :141-144 and, without ancestral
    alleles, against the f64 value ``1 - y`` :148-152; a site that matches the
    mirror is inverted (:156-160) even when it also matches ``y``.  Final
    condition :166.
    """
    if not (0 <= w <= 1):
        raise ValueError("Parameters w must be within the range [0, 1].")
    for op, y in y_list:
        if not (0 <= y <= 1):
            raise ValueError(f"Invalid value in y_list: {y}. within the range [0, 1].")
        if op not in _COMPARE:
            raise ValueError(
                f"Invalid operator in y_list: {op}. Must be '=', '<', '>', '<=', or '>='."
            )
    if len(src_gts_list) != len(y_list):
        raise ValueError("The length of src_gts_list and y_list must match.")

    ref_freq = allele_freq(ref_gts, ploidy[0])
    tgt_freq = allele_freq(tgt_gts, ploidy[1])
    src_freqs = [allele_freq(g, p) for g, p in zip(src_gts_list, ploidy[2:])]

    valid = _unit_interval(ref_freq) & _unit_interval(tgt_freq)
    for f in src_freqs:
        valid &= _unit_interval(f)

    hits_y = np.all([_COMPARE[op](f, y) for f, (op, y) in zip(src_freqs, y_list)], axis=0)
    if anc_allele_available:
        hits = hits_y
    else:
        hits_mirror = np.all(
            [_COMPARE[op](f, 1 - y) for f, (op, y) in zip(src_freqs, y_list)], axis=0
        )
        hits = hits_y | hits_mirror
        flip = hits_mirror & valid
        ref_freq[flip] = 1 - ref_freq[flip]
        tgt_freq[flip] = 1 - tgt_freq[flip]

    condition = valid & hits & (ref_freq < w)
    return ref_freq, tgt_freq, condition


# ---------------------------------------------------------------------------
# a3  UStatistic.compute                        sai/stats/u_statistic.py:37-99
# a4  QStatistic.compute                        sai/stats/q_statistic.py:37-104
# ---------------------------------------------------------------------------


def _require(kwargs: dict, names: Sequence[str]) -> None:
    # u_statistic.py:70-72, q_statistic.py:70-72
    missing = [k for k in names if k not in kwargs]
    if missing:
        raise ValueError(f"Missing required argument(s): {', '.join(missing)}")


def u_stat(
    ref_gts,
    tgt_gts,
    src_gts_list,
    ref_ploidy,
    tgt_ploidy,
    src_ploidy_list,
    **kwargs,
) -> dict[str, Any]:
    """U = number of sites with condition and tgt_freq > x (u_statistic.py:79-99)."""
    _require(kwargs, ["pos", "w", "x", "y_list", "anc_allele_available"])
    pos = kwargs["pos"]
    ploidy = [ref_ploidy, tgt_ploidy] + list(src_ploidy_list)
    _, tgt_freq, cond = matching_loci(
        ref_gts,
        tgt_gts,
        src_gts_list,
        kwargs["w"],
        kwargs["y_list"],
        ploidy,
        kwargs["anc_allele_available"],
    )
    cond &= tgt_freq > kwargs["x"]
    idx = np.where(cond)[0]
    return {"name": "U", "value": idx.size, "cdd_pos": pos[idx]}


def q_stat(
    ref_gts,
    tgt_gts,
    src_gts_list,
    ref_ploidy,
    tgt_ploidy,
    src_ploidy_list,
    **kwargs,
) -> dict[str, Any]:
    """Q = nanquantile of tgt_freq over condition sites (q_statistic.py:79-104).

    Empty selection -> NaN and ``np.array([])`` (:96-98); otherwise numpy's
    default 'linear' quantile (:100) and the positions with ``freq >= Q`` (:101).
    """
    _require(kwargs, ["pos", "w", "y_list", "anc_allele_available", "quantile"])
    pos = kwargs["pos"]
    ploidy = [ref_ploidy, tgt_ploidy] + list(src_ploidy_list)
    _, tgt_freq, cond = matching_loci(
        ref_gts,
        tgt_gts,
        src_gts_list,
        kwargs["w"],
        kwargs["y_list"],
        ploidy,
        kwargs["anc_allele_available"],
    )
    picked = tgt_freq[cond]
    picked_pos = pos[cond]
    if picked.size == 0:
        return {"name": "Q", "value": np.nan, "cdd_pos": np.array([])}
    thr = np.nanquantile(picked, kwargs["quantile"])
    return {"name": "Q", "value": thr, "cdd_pos": picked_pos[picked >= thr]}


def linear_quantile(sorted_vals: np.ndarray, q: float) -> float:
    """numpy's 'linear' quantile written out step by step (numpy 1.26/2.2
    ``_quantile`` + ``_get_indexes`` + ``_lerp``): virtual index
    ``(n-1)*q``; at or beyond the last index take the maximum; else
    ``a + (b-a)*g`` for ``g < 0.5`` and ``b - (b-a)*(1-g)`` otherwise.
    The HIP kernel follows exactly this sequence of f64 operations; the test
    suite checks this function bit-for-bit against ``np.nanquantile``.
    """
    n = len(sorted_vals)
    v = np.float64(n - 1) * np.float64(q)
    if v >= n - 1:
        return float(sorted_vals[n - 1])
    lo = int(np.floor(v))
    g = v - np.float64(lo)
    a = np.float64(sorted_vals[lo])
    b = np.float64(sorted_vals[lo + 1])
    d = b - a
    if g >= 0.5:
        return float(b - d * (np.float64(1) - g))
    return float(a + d * g)


# ---------------------------------------------------------------------------
# a9  split_genome                              sai/utils/utils.py:558-612
# ---------------------------------------------------------------------------


def split_windows(pos, window_size: int, step_size: int, start: Optional[int] = None):
    """Inclusive (start, end) windows.  Validation utils.py:593-598, first start
    :601, clamp :602-604, emission loop :607-610."""
    if step_size <= 0 or window_size <= 0:
        raise ValueError("`step_size` and `window_size` must be positive integers.")
    if step_size > window_size:
        raise ValueError("`step_size` cannot be greater than `window_size`.")
    if len(pos) == 0:
        raise ValueError("`pos` array must not be empty.")
    first = (pos[0] + step_size) // step_size * step_size - window_size + 1
    first = max(first, 1 if start is None else start)
    out = []
    s = first
    while s <= pos[-1]:
        out.append((s, s + window_size - 1))
        s += step_size
    return out


def split_window_ranges(windows: list, num_chunks: int) -> list:
    """Contiguous window ranges per chunk (chunk_generator.py:130-142): the first
    ``len % n`` chunks take one window more; a chunk is (first.start, last.end)."""
    base, extra = divmod(len(windows), num_chunks)
    out = []
    i = 0
    for c in range(num_chunks):
        j = i + base + (1 if c < extra else 0)
        if j > i:
            out.append((windows[i][0], windows[j - 1][1]))
        i = j
    return out


# ---------------------------------------------------------------------------
# a8  WindowGenerator._window_generator   sai/generators/window_generator.py
# ---------------------------------------------------------------------------


class Chrom:
    """POS + GT of one population (the oracle's ChromosomeData,
    sai/utils/genomic_dataclasses.py:25-46; REF/ALT are not needed here)."""

    __slots__ = ("POS", "GT")

    def __init__(self, POS: np.ndarray, GT: np.ndarray):
        self.POS = np.asarray(POS)
        self.GT = np.asarray(GT)


def target_windows(tgt_pos, win_len, win_step, start=None, end=None):
    """window_generator.py:132-144: without chunk bounds the grid comes from the
    target's positions, with bounds from ``[start, end - win_len + win_step]``."""
    if start is None and end is None:
        return split_windows(tgt_pos, win_len, win_step, start=None)
    return split_windows([start, end - win_len + win_step], win_len, win_step, start=start)


def iter_windows(
    chr_name,
    ref_data: dict[str, Chrom],
    tgt_data: dict[str, Chrom],
    src_data: dict[str, Chrom],
    win_len: int,
    win_step: int,
    ploidy_config=None,
    start=None,
    end=None,
    num_src: Optional[int] = None,
    out_data: Optional[dict[str, Chrom]] = None,
) -> Iterator[dict[str, Any]]:
    """The reference's per-window dicts, in its order, built its way: full-length
    inclusive position masks (:173-183), intersect1d chain (:193-197), empty
    window (:199-215), ``isin`` + ``compress`` per population (:217-231).
    Population order = dict order (product over ref, tgt, src combinations and
    outgroups, :162-166); the outgroup's positions join the intersection (:196-197)."""
    if num_src is None:
        num_src = len(src_data)
    src_combos = list(itertools.combinations(src_data.keys(), num_src))
    windows = {
        t: target_windows(tgt_data[t].POS, win_len, win_step, start, end) for t in tgt_data
    }
    out_pops = list(out_data) if out_data else [None]
    for ref_pop, tgt_pop, combo, out_pop in itertools.product(ref_data, tgt_data, src_combos, out_pops):
        r = ref_data[ref_pop]
        t = tgt_data[tgt_pop]
        srcs = [src_data[s] for s in combo]
        o = None if out_pop is None else out_data[out_pop]
        for w_start, w_end in windows[tgt_pop]:
            r_pos = r.POS[(r.POS >= w_start) & (r.POS <= w_end)]
            t_pos = t.POS[(t.POS >= w_start) & (t.POS <= w_end)]
            common = np.intersect1d(r_pos, t_pos)
            for s in srcs:
                common = np.intersect1d(common, s.POS[(s.POS >= w_start) & (s.POS <= w_end)])
            if o is not None:
                common = np.intersect1d(common, o.POS[(o.POS >= w_start) & (o.POS <= w_end)])
            item = {
                "chr_name": chr_name,
                "ref_pop": ref_pop,
                "tgt_pop": tgt_pop,
                "src_pop_list": combo,
                "out_pop": out_pop,
                "start": w_start,
                "end": w_end,
                "out_gts": None,
                "ploidy_config": ploidy_config,
            }
            if common.size == 0:
                item.update(pos=[], ref_gts=None, tgt_gts=None, src_gts_list=None)
            else:
                item.update(
                    pos=common,
                    ref_gts=r.GT.compress(np.isin(r.POS, common), axis=0),
                    tgt_gts=t.GT.compress(np.isin(t.POS, common), axis=0),
                    src_gts_list=[s.GT.compress(np.isin(s.POS, common), axis=0) for s in srcs],
                )
                if o is not None:
                    item["out_gts"] = o.GT.compress(np.isin(o.POS, common), axis=0)
            yield item


# ---------------------------------------------------------------------------
# a7  FeaturePreprocessor.run     sai/preprocessors/feature_preprocessor.py:63-191
# ---------------------------------------------------------------------------


def window_item(
    stat_params: dict[str, dict],
    ploidies: dict[str, dict[str, int]],
    anc_allele_available: bool,
    *,
    chr_name,
    ref_pop,
    tgt_pop,
    src_pop_list,
    out_pop,
    start,
    end,
    pos,
    ref_gts,
    tgt_gts,
    src_gts_list,
    out_gts=None,
    ploidy_config=None,
) -> dict[str, Any]:
    """One window's item dict.

    ``stat_params`` is the *validated* statistics mapping
    ({"U": {"ref": {pop: w}, "tgt": {pop: x}, "src": {pop: (op, y)}}, "Q": ...},
    stat_config.py:147-157) in YAML order; ``ploidies`` is the ploidy mapping
    (ploidy_config.py:66-96).  Item layout :116-129; the all-None window gives
    NaN and empty arrays (:131-144); thresholds are looked up per (ref_pop,
    tgt_pop) and the source thresholds/ploidies are taken positionally in config
    order (:152-185).  ``ploidy_config`` is only tested for None, as in the
    reference."""
    item = {
        "chr_name": chr_name,
        "start": start,
        "end": end,
        "ref_pop": ref_pop,
        "tgt_pop": tgt_pop,
        "src_pop_list": src_pop_list,
        "out_pop": "NA" if out_pop is None else out_pop,
        "nsnps": len(pos),
        "cdd_pos": {},
    }
    if ref_gts is None or tgt_gts is None or src_gts_list is None or ploidy_config is None:
        for name, prm in stat_params.items():
            if name in ("U", "Q"):
                item[name] = np.nan
                item["cdd_pos"][name] = np.array([])
            elif prm is True:  # feature_preprocessor.py:137-141
                item[name] = [np.nan] * len(src_pop_list) if len(src_pop_list) > 1 else np.nan
        return item
    four = None
    for name, prm in stat_params.items():
        if name not in ("U", "Q"):
            if prm is not True:  # feature_preprocessor.py:147-151
                continue
            if name == "DD":
                item[name] = dd_stat(ref_gts, tgt_gts, src_gts_list)
                continue
            if name not in ("fd", "df", "Danc", "Dplus"):
                raise ValueError(f"statistic {name} is outside the oracle")
            if four is None:
                out_ploidy = None if out_pop is None or "outgroup" not in ploidies else ploidies["outgroup"][out_pop]
                four = four_pop_stats(ref_gts, tgt_gts, src_gts_list, out_gts, ploidies["ref"][ref_pop],
                                      ploidies["tgt"][tgt_pop], list(ploidies["src"].values()), out_ploidy)  # fmt: skip
            item[name] = four[name]
            continue
        common = dict(
            ref_gts=ref_gts,
            tgt_gts=tgt_gts,
            src_gts_list=src_gts_list,
            ref_ploidy=ploidies["ref"][ref_pop],
            tgt_ploidy=ploidies["tgt"][tgt_pop],
            src_ploidy_list=list(ploidies["src"].values()),
            pos=pos,
            w=prm["ref"][ref_pop],
            y_list=list(prm["src"].values()),
            anc_allele_available=anc_allele_available,
        )
        if name == "U":
            res = u_stat(x=prm["tgt"][tgt_pop], **common)
        elif name == "Q":
            res = q_stat(quantile=prm["tgt"][tgt_pop], **common)
        else:
            raise ValueError(f"statistic {name} is outside the U/Q path")
        item["cdd_pos"][name] = res["cdd_pos"]
        item[name] = res["value"]
    return item


# ---------------------------------------------------------------------------
# a10 process_items + header      feature_preprocessor.py:193-258, sai.py:109-144
# ---------------------------------------------------------------------------


def header_line(stat_names: Sequence[str], src_pops: Sequence[str] = ()) -> str:
    """sai.py:109-131: U/Q are one column each; the other statistics one column per source
    population (``stat.src``) when there is more than one source."""
    cols = ["Chrom", "Start", "End", "Ref", "Tgt", "Src", "Outgroup", "N(Variants)"]
    for name in stat_names:
        if name in ("U", "Q") or len(src_pops) <= 1:
            cols.append(name)
        else:
            cols.extend(f"{name}.{sp}" for sp in src_pops)
    return "\t".join(cols) + "\n"


def log_header_line(key: str) -> str:
    """sai.py:139-144."""
    return f"Chrom\tStart\tEnd\t{key}_SNP\n"


def score_lines(items: Sequence[dict], stat_names: Sequence[str]) -> list[str]:
    """TSV rows (feature_preprocessor.py:206-237): values through ``str()``."""
    out = []
    for it in items:
        parts = []
        for s in stat_names:  # feature_preprocessor.py:217-228
            v = it.get(s)
            if isinstance(v, list) and len(v) == len(it["src_pop_list"]):
                parts.extend(str(x) for x in v)
            else:
                parts.append(str(v[0] if isinstance(v, list) and v else v))
        vals = "\t".join(parts)
        out.append(
            f"{it['chr_name']}\t{it['start']}\t{it['end']}\t{it['ref_pop']}\t{it['tgt_pop']}\t"
            f"{','.join(it['src_pop_list'])}\t{it['out_pop']}\t{it['nsnps']}\t{vals}\n"
        )
    return out


def log_lines(items: Sequence[dict], key: str) -> list[str]:
    """``.U.log`` / ``.Q.log`` rows (feature_preprocessor.py:241-258)."""
    out = []
    for it in items:
        cdd = it["cdd_pos"][key]
        txt = "NA" if cdd.size == 0 else ",".join(f"{it['chr_name']}:{p}" for p in cdd)
        out.append(f"{it['chr_name']}\t{it['start']}\t{it['end']}\t{txt}\n")
    return out


# ---------------------------------------------------------------------------
# chunk driver = ChunkPreprocessor.run on resident arrays
#                                   sai/preprocessors/chunk_preprocessor.py:105-147
# ---------------------------------------------------------------------------


def run_chunk(
    chr_name,
    ref_data,
    tgt_data,
    src_data,
    win_len,
    win_step,
    stat_params,
    ploidies,
    anc_allele_available,
    start=None,
    end=None,
    out_data=None,
) -> list[dict]:
    """All windows of one chunk through ``iter_windows`` + ``window_item``.  When a
    chunk is bounded the resident arrays are first cut to ``[start, end]`` the way
    the reference re-reads its VCF region (utils.py:118-121)."""
    if start is not None or end is not None:

        def cut(d):
            out = {}
            for k, c in d.items():
                m = (c.POS >= start) & (c.POS <= end)
                out[k] = Chrom(c.POS[m], c.GT[m])
            return out

        ref_data, tgt_data, src_data = cut(ref_data), cut(tgt_data), cut(src_data)
        out_data = cut(out_data) if out_data else out_data
    items = []
    for win in iter_windows(
        chr_name,
        ref_data,
        tgt_data,
        src_data,
        win_len,
        win_step,
        ploidy_config=ploidies,
        start=start,
        end=end,
        out_data=out_data,
    ):
        items.append(window_item(stat_params, ploidies, anc_allele_available, **win))
    return items


# ---------------------------------------------------------------------------
# ABBA-BABA family: fd, df, Danc, Dplus
#   calc_four_pops_freq / calc_pattern_sum        sai/stats/stat_utils.py:171-272
#   FdStatistic / DfStatistic / DancStatistic / DplusStatistic
#                                                 sai/stats/{fd,df,danc,dplus}_statistic.py
# ---------------------------------------------------------------------------


def four_pops_freq(ref_gts, tgt_gts, src_gts, out_gts=None, ref_ploidy=1, tgt_ploidy=1, src_ploidy=1, out_ploidy=1):
    """stat_utils.py:209-217: the outgroup frequency is 0 everywhere when there is no outgroup."""
    ref = allele_freq(ref_gts, ref_ploidy)
    out = np.zeros_like(ref) if out_gts is None else allele_freq(out_gts, out_ploidy)
    return ref, allele_freq(tgt_gts, tgt_ploidy), allele_freq(src_gts, src_ploidy), out


def pattern_sum(ref_freq, tgt_freq, src_freq, out_freq, pattern: str) -> float:
    """stat_utils.py:255-272: product over the four populations of f ('b') or 1 - f ('a'),
    multiplied in population order starting from 1.0, then np.sum (NaN sites poison the sum)."""
    if len(pattern) != 4:
        raise ValueError("Pattern must be a four-character string.")
    prod = np.ones_like(ref_freq)
    for f, c in zip((ref_freq, tgt_freq, src_freq, out_freq), pattern.lower()):
        if c == "a":
            prod *= 1 - f
        elif c == "b":
            prod *= f
        else:
            raise ValueError(f"Invalid character '{c}' in pattern. Only 'a' and 'b' allowed.")
    return float(np.sum(prod))


def _ratio(num: float, den: float) -> float:
    return num / den if den != 0 else np.nan


def four_pop_stats(ref_gts, tgt_gts, src_gts_list, out_gts, ref_ploidy, tgt_ploidy, src_ploidy_list, out_ploidy):
    """{"fd": [...], "df": [...], "Danc": [...], "Dplus": [...]}, one value per source
    (fd_statistic.py:61-89, df_statistic.py:60-84, danc_statistic.py:59-83,
    dplus_statistic.py:60-86)."""
    res = {"fd": [], "df": [], "Danc": [], "Dplus": []}
    for i in range(len(src_gts_list)):
        r, t, s, o = four_pops_freq(ref_gts, tgt_gts, src_gts_list[i], out_gts, ref_ploidy, tgt_ploidy,
                                    src_ploidy_list[i], out_ploidy)  # fmt: skip
        abba, baba = pattern_sum(r, t, s, o, "abba"), pattern_sum(r, t, s, o, "baba")
        bbaa, baaa, abaa = pattern_sum(r, t, s, o, "bbaa"), pattern_sum(r, t, s, o, "baaa"), pattern_sum(r, t, s, o, "abaa")
        d = np.maximum(t, s)
        abba_d, baba_d = pattern_sum(r, d, d, o, "abba"), pattern_sum(r, d, d, o, "baba")
        res["fd"].append(_ratio(abba - baba, abba_d - baba_d))
        res["df"].append(_ratio(abba - baba, abba + baba + 2 * bbaa))
        res["Danc"].append(_ratio(baaa - abaa, baaa + abaa))
        res["Dplus"].append(_ratio(abba - baba + baaa - abaa, abba + baba + baaa + abaa))
    return res


def numpy_pairwise_sum(a) -> float:
    """numpy's float add.reduce order for a contiguous 1-D array (numpy/_core/src/umath/
    loops_utils.h.src, *_pairwise_sum): plain loop below 8 elements; up to 128 elements eight
    running sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a tail loop; above that the
    array is halved (first half rounded down to a multiple of 8) recursively.  The HIP kernel
    sums in exactly this order; tests check this function bit-for-bit against np.sum."""
    a = np.asarray(a, dtype=np.float64)
    n = len(a)
    if n < 8:
        res = np.float64(0.0)
        for v in a:
            res = res + v
        return float(res)
    if n <= 128:
        r = [np.float64(a[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = r[j] + a[i + j]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + a[i]
            i += 1
        return float(res)
    n2 = n // 2
    n2 -= n2 % 8
    return float(np.float64(numpy_pairwise_sum(a[:n2])) + np.float64(numpy_pairwise_sum(a[n2:])))


def numpy_sum(a) -> float:
    """np.sum of a contiguous f64 array, operation by operation: the reduction runs over the
    ufunc buffer size (8192 elements) at a time, each piece pairwise-summed as above and added to
    the running total in order.  Checked bit-for-bit against np.sum in the tests."""
    a = np.asarray(a, dtype=np.float64)
    res = np.float64(0.0)
    for i in range(0, len(a), 8192):
        res = res + np.float64(numpy_pairwise_sum(a[i : i + 8192]))
    return float(res)


# ---------------------------------------------------------------------------
# DD                    sai/stats/dd_statistic.py:40-77
# ---------------------------------------------------------------------------


def dd_stat(ref_gts, tgt_gts, src_gts_list) -> list[float]:
    """Per source population: mean over its individuals of (mean city-block distance to the
    reference individuals - mean city-block distance to the target individuals), distances taken
    over the window's sites on the raw dosage values (missing calls enter as their negative
    numbers, as scipy's cdist sees them; dd_statistic.py:62-75)."""
    ref = np.asarray(ref_gts, dtype=np.float64)
    tgt = np.asarray(tgt_gts, dtype=np.float64)
    out = []
    for src in src_gts_list:
        s = np.asarray(src, dtype=np.float64)
        d_tgt = np.abs(s.T[:, None, :] - tgt.T[None, :, :]).sum(axis=2)  # == cdist(s.T, tgt.T, "cityblock")
        d_ref = np.abs(s.T[:, None, :] - ref.T[None, :, :]).sum(axis=2)
        out.append(np.mean(np.mean(d_ref, axis=1) - np.mean(d_tgt, axis=1)))
    return out
