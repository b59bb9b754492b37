#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports xin-huang/sai from /root/reference (read-only) with import stubs for
the three packages the image lacks (scikit-allel, natsort, pysam -- imported by
the reference but never executed on the U/Q path), feeds
it seeded inputs, and writes inputs + the reference's outputs as JSON.  Doubles
are stored as C99 hex strings so the fixtures are bit-exact.

Only data is written here: no reference source text ends up in the fixtures.
"""

from __future__ import annotations

import io
import json
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

REF = "/root/reference"
OUT = Path(__file__).resolve().parent


def _import_reference():
    allel = types.ModuleType("allel")
    allel.GenotypeVector = allel.GenotypeArray = object
    natsort = types.ModuleType("natsort")
    natsort.natsorted = sorted
    pysam = types.ModuleType("pysam")
    sys.modules.update(allel=allel, natsort=natsort, pysam=pysam)
    sys.path.insert(0, REF)
    import sai.stats  # noqa: F401  (registers U/Q)


def hx(v) -> str:
    v = float(v)
    return "nan" if v != v else v.hex()


def hxs(a) -> list[str]:
    return [hx(v) for v in np.asarray(a, dtype=np.float64).ravel()]


def ints(a) -> list:
    return np.asarray(a).astype(np.int64).tolist()


# ---------------------------------------------------------------------------
# 1. statistics: known-answer inputs of the reference's tests + seeded windows
# ---------------------------------------------------------------------------


def _known_answer_inputs():
    """Inputs (data only) of tests/stats/test_u_statistic.py and
    test_q_statistic.py, re-keyed; the expected outputs are produced by running
    the reference below and are asserted against the tests' published answers."""
    A = np.array
    cases = []

    def add(name, ref, tgt, srcs, pl, w, x, q, y, anc, expect=None):
        cases.append(
            dict(name=name, ref=A(ref), tgt=A(tgt), srcs=[A(s) for s in srcs], ploidy=pl,
                 pos=np.arange(len(ref)), w=w, x=x, quantile=q, y_list=y, anc=anc, expect=expect)
        )

    # test_u_statistic.py:26-73
    r, t, s = [[0, 0, 1], [0, 0, 0], [1, 1, 1]], [[1, 1, 1], [1, 0, 0], [0, 1, 0]], [[0, 0, 0], [1, 1, 1], [1, 0, 1]]
    add("u_basic", r, t, [s], [1, 1, 1], 0.5, 0.5, 0.95, [("=", 0)], False, dict(U=1, U_pos=[0]))
    add("u_basic_anc", r, t, [s], [1, 1, 1], 0.5, 0.5, 0.95, [("=", 1)], True, dict(U=0, U_pos=[]))
    # :76-108
    add("u_no_match", [[0, 1, 1], [1, 1, 1]], [[0, 0, 0], [1, 0, 1]], [[[1, 1, 1], [1, 1, 1]]],
        [1, 1, 1], 0.3, 0.5, 0.95, [("=", 0)], False, dict(U=0, U_pos=[]))
    # :111-143
    add("u_all_match", [[0, 0, 0], [0, 0, 0]], [[1, 1, 1], [1, 1, 1]], [[[0, 0, 0], [0, 0, 0]]],
        [1, 1, 1], 0.5, 0.5, 0.95, [("=", 0)], False, dict(U=2, U_pos=[0, 1]))
    # :146-178
    s2 = [[1, 1, 1], [0, 1, 1], [1, 1, 1]]
    add("u_two_sources", [[0, 0, 1], [0, 0, 0], [1, 1, 1]], [[0, 1, 1], [0, 0, 1], [1, 1, 1]], [s2, s2],
        [1, 1, 1, 1], 0.5, 0.5, 0.95, [("=", 1.0), ("=", 1.0)], False, dict(U=1, U_pos=[0]))
    # :181-209
    add("u_mixed_ploidy", [[0, 1, 0], [0, 1, 0], [2, 1, 0]], [[1, 1, 0], [1, 1, 1], [1, 1, 1]],
        [[[0, 0, 0], [1, 1, 1], [0, 0, 0]]], [3, 1, 2], 0.5, 0.5, 0.95, [("=", 0)], False, dict(U=2, U_pos=[0, 2]))
    # test_q_statistic.py:26-72
    rq, tq, sq = [[0, 0, 1], [0, 0, 0], [1, 1, 1]], [[0, 1, 1], [0, 0, 1], [1, 1, 1]], [[1, 1, 1], [0, 1, 1], [1, 1, 1]]
    add("q_basic", rq, tq, [sq], [1, 1, 1], 0.5, 0.5, 0.95, [("=", 1.0)], False, dict(Q=0.66667, Q_pos=[0]))
    add("q_basic_anc", rq, tq, [sq], [1, 1, 1], 0.5, 0.5, 0.95, [("=", 1.0)], True, dict(Q=0.66667, Q_pos=[0]))
    # :75-108
    add("q_no_match", [[0, 0, 1], [0, 0, 0]], [[0, 1, 1], [1, 1, 1]], [[[1, 1, 1], [1, 1, 1]]],
        [1, 1, 1], 0.3, 0.5, 0.95, [("=", 0.0)], False, dict(Q=None, Q_pos=[]))
    # :111-145
    add("q_median", [[0, 0, 1], [1, 0, 0], [0, 0, 1]], [[0, 1, 1], [1, 1, 1], [1, 1, 1]],
        [[[0, 0, 0], [1, 1, 1], [1, 1, 1]]], [1, 1, 1], 0.5, 0.5, 0.5, [("=", 1.0)], False, dict(Q=1.0, Q_pos=[1, 2]))
    # :148-180
    add("q_edge", [[0, 0, 1], [0, 0, 0], [1, 1, 1]], [[0, 1, 1], [1, 1, 1], [0, 0, 0]],
        [[[0, 0, 0], [1, 1, 1], [1, 1, 1]]], [1, 1, 1], 0.95, 0.5, 0.95, [("=", 1.0)], False,
        dict(Q=0.9666666666666667, Q_pos=[1]))
    # :183-213 and :216-243
    r4 = [[1, 1, 0], [0, 1, 1], [1, 1, 1], [0, 0, 1]]
    t4 = [[0, 0, 0], [1, 1, 1], [1, 1, 1], [1, 1, 1]]
    sa = [[0, 0, 0], [1, 1, 1], [1, 1, 1], [0, 0, 1]]
    sb = [[1, 1, 1], [1, 1, 1], [0, 0, 0], [1, 1, 1]]
    add("q_two_sources", r4, t4, [sa, sb], [1, 1, 1, 1], 0.5, 0.5, 0.95, [("=", 1), ("=", 1)], False, dict(Q=None, Q_pos=[]))
    add("q_mixed_ploidy", r4, t4, [sa, sb], [2, 2, 4, 4], 0.5, 0.5, 0.95, [("=", 1), ("=", 1)], False, dict(Q=None, Q_pos=[]))
    # test_stat_utils.py:115-141 (all five operators, two sources)
    rm, tm = [[0, 1, 0], [1, 1, 0], [0, 0, 1]], [[1, 1, 0], [0, 1, 1], [1, 1, 1]]
    sm = [[[0, 0, 1], [1, 1, 0], [0, 1, 1]], [[1, 1, 0], [1, 0, 0], [1, 1, 0]]]
    for i, y in enumerate([("=", 0.5), ("<", 0.4), (">", 0.3), ("<=", 0.6), (">=", 0.2)]):
        add(f"ops_{i}", rm, tm, sm, [2, 2, 2], 0.5, 0.5, 0.5, [y, y], False)
    return cases


def _float_trap_inputs():
    """SURVEY.md section 8c semantics 1-4."""
    A = np.array
    cases = []
    # 1. fl(1/100) is not < 0.01: 50 diploids, one ALT allele
    ref = np.zeros((3, 50), dtype=np.int64)
    ref[0, 0] = 1  # 1/100
    ref[1, 0] = 0  # 0
    ref[2, :2] = 1  # 2/100
    tgt = np.full((3, 4), 2)
    src = np.full((3, 1), 2)
    cases.append(dict(name="trap_w_equal", ref=ref, tgt=tgt, srcs=[src], ploidy=[2, 2, 2], pos=A([10, 20, 30]),
                      w=0.01, x=0.5, quantile=0.95, y_list=[("=", 1.0)], anc=True, expect=None))
    # 2. ("=", 0.9) mirror 1-0.9 != 1/10: 5 diploid sources
    src5 = np.zeros((3, 5), dtype=np.int64)
    src5[0, :] = [2, 2, 2, 2, 1]  # 9/10
    src5[1, 0] = 1  # 1/10
    src5[2, :] = 2
    cases.append(dict(name="trap_mirror_09", ref=np.zeros((3, 4), dtype=np.int64), tgt=np.full((3, 4), 2), srcs=[src5],
                      ploidy=[2, 2, 2], pos=A([1, 2, 3]), w=0.5, x=0.2, quantile=0.5, y_list=[("=", 0.9)], anc=False, expect=None))
    # 3. both y and 1-y match -> inverted
    srch = A([[1], [1], [2], [0]])
    refh = A([[0, 0, 0, 0], [2, 2, 2, 2], [0, 0, 0, 1], [2, 2, 2, 1]])
    tgth = A([[2, 2, 2, 2], [0, 0, 0, 0], [2, 2, 1, 1], [0, 0, 1, 1]])
    for i, y in enumerate([("=", 0.5), (">=", 0.2), ("<=", 0.8), (">", 0.0), ("<", 1.0)]):
        cases.append(dict(name=f"trap_both_{i}", ref=refh, tgt=tgth, srcs=[srch], ploidy=[2, 2, 2], pos=A([5, 6, 7, 8]),
                          w=0.3, x=0.4, quantile=0.75, y_list=[y], anc=False, expect=None))
    # 4. all-missing population at a site, half-missing calls, dosage above ploidy
    refm = A([[-2, -2, -2], [0, -2, 0], [0, 0, 0], [0, 0, 0], [-1, -1, 0]])
    tgtm = A([[2, 2, 2], [2, 2, -2], [-2, -2, -2], [2, 3, 2], [2, -1, 2]])
    srcm = A([[2], [2], [2], [2], [-2]])
    for anc in (True, False):
        cases.append(dict(name=f"trap_missing_{int(anc)}", ref=refm, tgt=tgtm, srcs=[srcm], ploidy=[2, 2, 2],
                          pos=A([100, 200, 300, 400, 500]), w=0.5, x=0.5, quantile=0.5, y_list=[("=", 1.0)], anc=anc, expect=None))
    return cases


def _random_inputs(n_cases=48, seed=20260630):
    rng = np.random.default_rng(seed)
    ops = ["=", "<", ">", "<=", ">="]
    cases = []
    for c in range(n_cases):
        n_sites = int(rng.integers(1, 60))
        n_src = int(rng.integers(1, 4))
        pl = [int(rng.integers(1, 5)) for _ in range(2 + n_src)]
        sizes = [int(rng.integers(1, 14)), int(rng.integers(1, 14))] + [int(rng.integers(1, 4)) for _ in range(n_src)]
        miss = float(rng.choice([0.0, 0.02, 0.2]))
        p_site = rng.random(n_site := n_sites) ** 2

        def block(n_ind, ploidy, flip=False, fix=0.0):
            p = np.where(rng.random(n_site) < fix, np.round(p_site), p_site)
            g = rng.binomial(ploidy, np.broadcast_to(p[:, None], (n_site, n_ind))).astype(np.int64)
            m = rng.random((n_site, n_ind)) < miss
            g[m] = -rng.integers(1, ploidy + 1, size=int(m.sum()))
            return g

        ref = block(sizes[0], pl[0])
        tgt = block(sizes[1], pl[1])
        srcs = [block(sizes[2 + k], pl[2 + k], fix=0.7) for k in range(n_src)]
        pos = np.sort(rng.choice(np.arange(1, 100000), size=n_sites, replace=False))
        grid = [0.0, 0.1, 0.2, 0.25, 1 / 3, 0.5, 0.75, 0.8, 0.9, 1.0]
        y_list = [(str(rng.choice(ops)), float(rng.choice(grid))) for _ in range(n_src)]
        if c % 3 == 0:
            y_list = [("=", float(rng.choice([0.0, 0.5, 1.0]))) for _ in range(n_src)]
        cases.append(dict(name=f"rand_{c}", ref=ref, tgt=tgt, srcs=srcs, ploidy=pl, pos=pos,
                          w=float(rng.choice([0.01, 0.1, 0.3, 0.5, 1.0])), x=float(rng.choice([0.0, 0.2, 0.5, 0.8])),
                          quantile=float(rng.choice([0.0, 0.25, 0.5, 0.9, 0.95, 1.0])), y_list=y_list,
                          anc=bool(c % 2), expect=None))
    return cases


def make_stats():
    from sai.stats import QStatistic, UStatistic, calc_freq, compute_matching_loci

    out = []
    for c in _known_answer_inputs() + _float_trap_inputs() + _random_inputs():
        kw = dict(ref_gts=c["ref"], tgt_gts=c["tgt"], src_gts_list=c["srcs"], ref_ploidy=c["ploidy"][0],
                  tgt_ploidy=c["ploidy"][1], src_ploidy_list=c["ploidy"][2:])
        u = UStatistic(**kw).compute(pos=c["pos"], w=c["w"], x=c["x"], y_list=c["y_list"], anc_allele_available=c["anc"])
        q = QStatistic(**kw).compute(pos=c["pos"], w=c["w"], y_list=c["y_list"], quantile=c["quantile"],
                                     anc_allele_available=c["anc"])
        rf, tf, cond = compute_matching_loci(c["ref"], c["tgt"], c["srcs"], c["w"], c["y_list"], c["ploidy"], c["anc"])
        raw = [calc_freq(c["ref"], c["ploidy"][0]), calc_freq(c["tgt"], c["ploidy"][1])] + [
            calc_freq(s, p) for s, p in zip(c["srcs"], c["ploidy"][2:])
        ]
        e = c["expect"]
        if e:  # the reference's own published answers still hold for what we captured
            if "U" in e:
                assert u["value"] == e["U"] and ints(u["cdd_pos"]) == e["U_pos"], c["name"]
            if "Q" in e:
                if e["Q"] is None:
                    assert np.isnan(q["value"]) and q["cdd_pos"].size == 0, c["name"]
                else:
                    assert np.isclose(q["value"], e["Q"]) and ints(q["cdd_pos"]) == e["Q_pos"], c["name"]
        assert isinstance(u["value"], int)
        out.append(dict(
            name=c["name"], ref_gts=ints(c["ref"]), tgt_gts=ints(c["tgt"]), src_gts_list=[ints(s) for s in c["srcs"]],
            ploidy=c["ploidy"], pos=ints(c["pos"]), w=hx(c["w"]), x=hx(c["x"]), quantile=hx(c["quantile"]),
            y_list=[[op, hx(y)] for op, y in c["y_list"]], anc_allele_available=c["anc"],
            out=dict(U=u["value"], U_cdd_pos=ints(u["cdd_pos"]), Q=hx(q["value"]), Q_cdd_pos=ints(q["cdd_pos"]),
                     Q_cdd_dtype=str(q["cdd_pos"].dtype), raw_freq=[hxs(f) for f in raw], ref_freq=hxs(rf),
                     tgt_freq=hxs(tf), condition=[bool(b) for b in cond]),
        ))
    (OUT / "stats_cases.json").write_text(json.dumps(out, separators=(",", ":")) + "\n")
    print("stats_cases.json", len(out))


# ---------------------------------------------------------------------------
# 2. calc_freq known answers + error behaviour
# ---------------------------------------------------------------------------


def make_errors():
    from sai.stats import QStatistic, UStatistic, calc_freq, compute_matching_loci

    A = np.array
    ref, tgt = A([[0, 1, 0], [1, 1, 0], [0, 0, 1]]), A([[1, 1, 0], [0, 1, 1], [1, 1, 1]])
    srcs = [A([[0, 0, 1], [1, 1, 0], [0, 1, 1]]), A([[1, 1, 0], [1, 0, 0], [1, 1, 0]])]
    y2 = [("=", 0.5), ("=", 0.5)]
    rec = []

    def capture(label, fn):
        try:
            fn()
            rec.append(dict(label=label, exc=None, msg=None))
        except Exception as e:  # noqa: BLE001
            rec.append(dict(label=label, exc=type(e).__name__, msg=str(e)))

    capture("w_low", lambda: compute_matching_loci(ref, tgt, srcs, -0.1, y2, [2, 2, 2], False))
    capture("w_high", lambda: compute_matching_loci(ref, tgt, srcs, 1.1, y2, [2, 2, 2], False))
    capture("y_low", lambda: compute_matching_loci(ref, tgt, srcs, 0.5, [("=", -0.1)], [2, 2, 2], False))
    capture("y_high", lambda: compute_matching_loci(ref, tgt, srcs, 0.5, [("=", 1.1)], [2, 2, 2], False))
    capture("bad_op", lambda: compute_matching_loci(ref, tgt, srcs, 0.5, [("invalid", 0.5)], [2, 2, 2], False))
    capture("len_mismatch", lambda: compute_matching_loci(ref, tgt, srcs, 0.5, [("=", 0.5)], [2, 2, 2], False))
    capture("ploidy_none", lambda: calc_freq(ref, ploidy=None))
    capture("ploidy_float", lambda: calc_freq(ref, ploidy=9.9))
    capture("ploidy_neg", lambda: calc_freq(ref, ploidy=-100))
    kw = dict(ref_gts=ref, tgt_gts=tgt, src_gts_list=srcs[:1], ref_ploidy=3, tgt_ploidy=1, src_ploidy_list=[2])
    capture("u_missing_kw", lambda: UStatistic(**kw).compute(pos=A([0, 1, 2]), w=0.5, x=0.5, y_list=[("=", 0)]))
    capture("q_missing_kw", lambda: QStatistic(**kw).compute(pos=A([0, 1, 2]), w=0.5, quantile=0.95,
                                                            anc_allele_available=False))
    freq = []
    for gts, pl in [([[1, 0, 0, 1], [0, 0, 0, 0], [1, 1, 1, 1]], 1), ([[1, -1, -1, 1], [-1, -1, -1, -1], [1, -1, 1, 1]], 1),
                    ([[1, 1], [0, 0], [2, 2]], 2), ([[1, -1], [0, 0], [-2, 2]], 2), ([[1, 2, 3], [0, 0, 0], [3, 3, 3]], 3),
                    ([[2, 2, 2, 2], [1, 3, 0, 4], [0, 0, 0, 0]], 4)]:
        freq.append(dict(gts=gts, ploidy=pl, freq=hxs(calc_freq(A(gts), ploidy=pl))))
    (OUT / "errors_and_freq.json").write_text(json.dumps(dict(errors=rec, calc_freq=freq), separators=(",", ":")) + "\n")
    print("errors_and_freq.json", len(rec), len(freq))


# ---------------------------------------------------------------------------
# 3. split_genome + chunk ranges
# ---------------------------------------------------------------------------


def make_windows_grid():
    from sai.generators.chunk_generator import ChunkGenerator
    from sai.utils import split_genome

    cases = []
    for pos, win, step, start in [
        ([0, 100], 30, 20, None),  # tests/utils/test_utils.py:423-431
        ([2309, 48989], 10000, 5000, None),  # tests/generators/test_chunk_generator.py
        ([2309, 48989], 1000, 500, None),  # 95 windows (380 = 95 x 4 combos)
        ([111, 6666], 6666, 6666, None),
        ([5, 12], 50, 10, None),  # clamp to 1
        ([25001, 25001 + 55000 - 30000 - 10000 + 5000], 10000, 5000, 25001),  # chunk re-derivation
        ([1, 30000 - 10000 + 5000], 10000, 5000, 1),
        ([1000, 250000000], 50000, 25000, None),
        ([40, 40], 7, 7, None),
        ([9999, 10000], 50000, 10000, None),
        ([17, 1000], 100, 100, 20),
    ]:
        w = split_genome(np.array(pos), win, step, start)
        cases.append(dict(pos=pos, window_size=win, step_size=step, start=start, n=len(w),
                          head=[list(map(int, t)) for t in w[:6]], tail=[list(map(int, t)) for t in w[-3:]]))
    errs = []
    for pos, win, step in [([0, 10], 20, 25), ([], 30, 10), ([1, 2], 0, 1), ([1, 2], 5, 0)]:
        try:
            split_genome(np.array(pos), win, step)
            errs.append(dict(pos=pos, window_size=win, step_size=step, msg=None))
        except ValueError as e:
            errs.append(dict(pos=pos, window_size=win, step_size=step, msg=str(e)))
    cg = object.__new__(ChunkGenerator)
    chunks = []
    for pos, win, step, n in [([2309, 48989], 10000, 5000, 2), ([2309, 48989], 10000, 5000, 3),
                              ([2309, 48989], 1000, 500, 8), ([2309, 48989], 10000, 5000, 40), ([1, 100], 50, 50, 1)]:
        w = split_genome(np.array(pos), win, step)
        chunks.append(dict(pos=pos, window_size=win, step_size=step, num_chunks=n,
                           chunks=[list(map(int, c)) for c in cg._split_windows_ranges(w, n)]))
    assert chunks[0]["chunks"] == [[1, 30000], [25001, 55000]]
    (OUT / "window_grid.json").write_text(json.dumps(dict(split=cases, errors=errs, chunks=chunks), separators=(",", ":")) + "\n")
    print("window_grid.json", len(cases), len(chunks))


# ---------------------------------------------------------------------------
# 4. WindowGenerator + FeaturePreprocessor + process_items on seeded chromosomes
# ---------------------------------------------------------------------------


def _seeded_chromosome(seed, n_sites, pops, miss):
    """pops: list of (group, name, n_ind, ploidy).  Returns pos and {group: {name: GT int64}}."""
    rng = np.random.default_rng(seed)
    pos = np.cumsum(rng.integers(1, 60, size=n_sites)).astype(np.int32) + 1000
    p = rng.random(n_sites) ** 3
    intro = rng.random(n_sites) < 0.03
    data = {"ref": {}, "tgt": {}, "src": {}}
    for group, name, n_ind, ploidy in pops:
        pp = p.copy()
        if group == "ref":
            pp[intro] = 0.0
        elif group == "tgt":
            pp[intro] = 0.2 + 0.7 * rng.random(int(intro.sum()))
        else:
            pp[intro] = 1.0
        g = rng.binomial(ploidy, np.broadcast_to(pp[:, None], (n_sites, n_ind))).astype(np.int64)
        m = rng.random((n_sites, n_ind)) < miss
        g[m] = -ploidy
        data[group][name] = g
    return pos, data


def make_pipeline():
    from sai.configs import PloidyConfig, StatConfig
    from sai.generators import WindowGenerator
    from sai.preprocessors import FeaturePreprocessor
    from sai.utils import split_genome
    from sai.utils.genomic_dataclasses import ChromosomeData
    from itertools import combinations

    scenarios = [
        dict(name="one_src", seed=11, n_sites=1500, miss=0.01, win_len=5000, win_step=2500, start=None, end=None, anc=False,
             pops=[("ref", "AFR", 12, 2), ("tgt", "CHB", 10, 2), ("src", "Nean", 1, 2)],
             stats={"U": {"ref": {"AFR": 0.1}, "tgt": {"CHB": 0.2}, "src": {"Nean": "=1"}},
                    "Q": {"ref": {"AFR": 0.1}, "tgt": {"CHB": 0.95}, "src": {"Nean": "=1"}}}),
        dict(name="two_tgt_two_src_chunk", seed=12, n_sites=1200, miss=0.02, win_len=4000, win_step=1000, start=9001, end=40000,
             anc=True, pops=[("ref", "r1", 9, 2), ("tgt", "t1", 7, 2), ("tgt", "t2", 5, 4), ("src", "s1", 1, 2), ("src", "s2", 2, 1)],
             stats={"Q": {"ref": {"r1": 0.3}, "tgt": {"t1": 0.9, "t2": 0.5}, "src": {"s1": ">=0.5", "s2": "=1"}},
                    "U": {"ref": {"r1": 0.2}, "tgt": {"t1": 0.3, "t2": 0.1}, "src": {"s1": "=1", "s2": ">0.4"}}}),
        dict(name="gappy", seed=13, n_sites=900, miss=0.0, win_len=3000, win_step=3000, start=None, end=None, anc=False,
             pops=[("ref", "A", 6, 2), ("tgt", "B", 6, 2), ("src", "N", 2, 2)],
             stats={"U": {"ref": {"A": 0.01}, "tgt": {"B": 0.5}, "src": {"N": "=1"}}}),
    ]
    out = []
    for sc in scenarios:
        pos, data = _seeded_chromosome(sc["seed"], sc["n_sites"], sc["pops"], sc["miss"])
        if sc["name"] == "gappy":  # open a hole so some windows are empty
            keep = ~((pos > 7000) & (pos < 14500))
            pos = pos[keep]
            data = {g: {k: v[keep] for k, v in d.items()} for g, d in data.items()}
        start, end = sc["start"], sc["end"]
        sel = np.ones(len(pos), bool) if start is None else (pos >= start) & (pos <= end)
        ploidies = {"ref": {}, "tgt": {}, "src": {}}
        for g, name, _, pl in sc["pops"]:
            ploidies[g][name] = pl
        pc = PloidyConfig(ploidies)
        wg = object.__new__(WindowGenerator)
        wg.win_len, wg.win_step, wg.chr_name, wg.ploidy_config = sc["win_len"], sc["win_step"], "21", pc
        for g in ("ref", "tgt", "src"):
            setattr(wg, f"{g}_data", {k: ChromosomeData(POS=pos[sel].copy(), REF=None, ALT=None, GT=v[sel].copy())
                                      for k, v in data[g].items()})
            setattr(wg, f"{g}_samples", {k: [f"{k}_{i}" for i in range(v.shape[1])] for k, v in data[g].items()})
        wg.out_data = wg.out_samples = None
        wg.num_src = len(data["src"])
        wg.src_combinations = list(combinations(wg.src_samples.keys(), wg.num_src))
        wg.tgt_windows = {
            t: split_genome(pos=(wg.tgt_data[t].POS if start is None and end is None
                                 else [start, end - sc["win_len"] + sc["win_step"]]),
                            window_size=sc["win_len"], step_size=sc["win_step"], start=start)
            for t in wg.tgt_samples
        }
        stat_config = StatConfig(json.loads(json.dumps(sc["stats"])))
        with tempfile.TemporaryDirectory() as td:
            tsv = os.path.join(td, "o.tsv")
            fp = FeaturePreprocessor(output_file=tsv, stat_config=stat_config, anc_allele_available=sc["anc"])
            items, wins = [], []
            for item in wg.get():
                p = item["pos"]
                lo = int(np.searchsorted(pos, p[0])) if len(p) else -1
                if len(p):
                    assert np.array_equal(pos[lo:lo + len(p)], p)
                wins.append([item["ref_pop"], item["tgt_pop"], list(item["src_pop_list"]), int(item["start"]),
                             int(item["end"]), len(p), lo])
                items.extend(fp.run(**item))
            fp.process_items(items)
            text = {"tsv": open(tsv).read()}
            for k in ("U", "Q"):
                f = os.path.join(td, f"o.{k}.log")
                if os.path.exists(f):
                    text[k] = open(f).read()
        out.append(dict(
            name=sc["name"], chr_name="21", win_len=sc["win_len"], win_step=sc["win_step"], start=start, end=end,
            anc_allele_available=sc["anc"], stats=sc["stats"], ploidies=ploidies, pos=ints(pos),
            gts={g: {k: ints(v) for k, v in d.items()} for g, d in data.items()}, windows=wins,
            items=[dict(U=(None if "U" not in it else (hx(it["U"]) if isinstance(it["U"], float) else int(it["U"]))),
                        Q=(None if "Q" not in it else hx(it["Q"])),
                        U_cdd=(ints(it["cdd_pos"]["U"]) if "U" in it["cdd_pos"] else None),
                        Q_cdd=(ints(it["cdd_pos"]["Q"]) if "Q" in it["cdd_pos"] else None),
                        nsnps=int(it["nsnps"])) for it in items],
            text=text,
        ))
        print(sc["name"], len(wins), "windows")
    (OUT / "pipeline.json").write_text(json.dumps(out, separators=(",", ":")) + "\n")

    # FeaturePreprocessor on inline arrays (tests/preprocessors/test_feature_preprocessor.py:54-159)
    A = np.array
    sc = StatConfig({"DD": False,
                     "U": {"ref": {"ref1": 0.3}, "tgt": {"tgt1": 0.5}, "src": {"src1": "=1", "src2": "=1"}},
                     "Q": {"ref": {"ref1": 0.3}, "tgt": {"tgt1": 0.95}, "src": {"src1": "=0.2", "src2": "=0.4"}}})
    with tempfile.TemporaryDirectory() as td:
        tsv = os.path.join(td, "t.tsv")
        fp = FeaturePreprocessor(output_file=tsv, stat_config=sc)
        kw = dict(chr_name="21", ref_pop="ref1", tgt_pop="tgt1", src_pop_list=["src1", "src2"], out_pop=None, start=1000,
                  end=2000, pos=A([100, 200, 300]), ref_gts=A([[0, 0, 1], [1, 1, 0], [0, 1, 1]]),
                  tgt_gts=A([[0, 1, 1], [1, 1, 1], [0, 0, 1]]),
                  src_gts_list=[A([[0, 0, 0], [1, 0, 0], [1, 1, 1]]), A([[1, 1, 1], [0, 1, 1], [0, 0, 1]])], out_gts=None,
                  ploidy_config=PloidyConfig({"ref": {"ref1": 1}, "tgt": {"tgt1": 1}, "src": {"src1": 1}}))
        full = fp.run(**kw)[0]
        kw2 = dict(kw, ref_gts=None, tgt_gts=None, src_gts_list=None, ploidy_config=None)
        none = fp.run(**kw2)[0]
        item = {"chr_name": "21", "start": 1000, "end": 2000, "ref_pop": "ref1", "tgt_pop": "tgt1", "out_pop": "NA",
                "src_pop_list": ["src1", "src2"], "nsnps": 10, "U": 5, "Q": 0.8,
                "cdd_pos": {"U": A([]), "Q": A([])}}
        fp.process_items([item])
        line = open(tsv).read()
        assert line == "21\t1000\t2000\tref1\ttgt1\tsrc1,src2\tNA\t10\t5\t0.8\n"
    inline = dict(
        full=dict(U=int(full["U"]), Q=hx(full["Q"]), U_cdd=ints(full["cdd_pos"]["U"]), Q_cdd=ints(full["cdd_pos"]["Q"]),
                  nsnps=full["nsnps"], out_pop=full["out_pop"], has_DD=("DD" in full)),
        none=dict(U=hx(none["U"]), Q=hx(none["Q"]), U_cdd=ints(none["cdd_pos"]["U"]), Q_cdd=ints(none["cdd_pos"]["Q"]),
                  nsnps=none["nsnps"]),
        process_items_line=line,
    )
    (OUT / "feature_inline.json").write_text(json.dumps(inline, separators=(",", ":")) + "\n")
    print("feature_inline.json")


# ---------------------------------------------------------------------------
# 5. tiny VCF (tests/data/example.vcf genotypes, typed in as arrays by our own
#    reader in the build; here: through the reference's stats from in-memory GT)
# ---------------------------------------------------------------------------


def make_example_vcf():
    """Expected `sai score` text for the 15-site example VCF.  The reference's VCF
    reader (scikit-allel) is not available, so the genotype block is parsed here
    by a 10-line splitter and everything downstream is the reference."""
    from sai.configs import PloidyConfig, StatConfig
    from sai.preprocessors import FeaturePreprocessor

    vcf = Path(REF) / "tests" / "data" / "example.vcf"
    pos, rows = [], []
    for ln in vcf.read_text().splitlines():
        if ln.startswith("#"):
            continue
        f = ln.split("\t")
        pos.append(int(f[1]))
        rows.append([sum(-1 if a == "." else int(a) for a in gt.replace("/", "|").split("|")) for gt in f[9:]])
    gt = np.array(rows, dtype=np.int64)
    pos = np.array(pos, dtype=np.int32)
    ref, tgt, src = gt[:, 0:5], gt[:, 5:10], gt[:, 10:11]
    res = {}
    for label, stats in [("q_only", {"Q": {"ref": {"AFR": 0.3}, "tgt": {"CHB": 0.95}, "src": {"Nean": "=1"}}}),
                         ("u_and_q", {"U": {"ref": {"AFR": 0.3}, "tgt": {"CHB": 0.5}, "src": {"Nean": "=1"}},
                                      "Q": {"ref": {"AFR": 0.3}, "tgt": {"CHB": 0.95}, "src": {"Nean": "=1"}}}),
                         ("example_config", {"U": {"ref": {"AFR": 0.01}, "tgt": {"CHB": 0.5}, "src": {"Nean": "=1"}},
                                             "Q": {"ref": {"AFR": 0.3}, "tgt": {"CHB": 0.95}, "src": {"Nean": "=1"}}})]:
        with tempfile.TemporaryDirectory() as td:
            tsv = os.path.join(td, "o.tsv")
            fp = FeaturePreprocessor(output_file=tsv, stat_config=StatConfig(json.loads(json.dumps(stats))))
            items = fp.run(chr_name="21", ref_pop="AFR", tgt_pop="CHB", src_pop_list=("Nean",), out_pop=None, start=1, end=6666,
                           pos=pos, ref_gts=ref, tgt_gts=tgt, src_gts_list=[src], out_gts=None,
                           ploidy_config=PloidyConfig({"ref": {"AFR": 2}, "tgt": {"CHB": 2}, "src": {"Nean": 2}}))
            fp.process_items(items)
            text = {"tsv": open(tsv).read()}
            for k in ("U", "Q"):
                f = os.path.join(td, f"o.{k}.log")
                if os.path.exists(f):
                    text[k] = open(f).read()
        res[label] = dict(stats=stats, text=text)
    assert res["q_only"]["text"]["tsv"] == "21\t1\t6666\tAFR\tCHB\tNean\tNA\t15\t0.9\n"  # tests/test_sai.py:63
    assert res["u_and_q"]["text"]["tsv"] == "21\t1\t6666\tAFR\tCHB\tNean\tNA\t15\t3\t0.9\n"  # test_feature_preprocessor.py:223
    res["genotypes"] = dict(pos=ints(pos), gt=ints(gt))
    (OUT / "example_vcf.json").write_text(json.dumps(res, separators=(",", ":")) + "\n")
    print("example_vcf.json")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present: fixtures can only be regenerated in the build container")
    _import_reference()
    make_stats()
    make_errors()
    make_windows_grid()
    make_pipeline()
    make_example_vcf()


# ---------------------------------------------------------------------------
# 6. `sai outlier` (sai/sai.py:154-230) on single-chromosome score tables, where the
#    `natsorted = sorted` import stub orders rows exactly like natsort would
# ---------------------------------------------------------------------------


def make_outlier():
    import warnings

    from sai.sai import outlier

    tables = {"test.q.scores": (Path(REF) / "tests" / "data" / "test.q.scores").read_text()}
    for sc in json.loads((OUT / "pipeline.json").read_text()):
        names = list(sc["stats"].keys())
        head = "Chrom\tStart\tEnd\tRef\tTgt\tSrc\tOutgroup\tN(Variants)\t" + "\t".join(names) + "\n"
        tables[sc["name"]] = head + sc["text"]["tsv"]
    out = []
    for name, text in tables.items():
        for quantile in (0.25, 0.5, 0.75, 0.99):
            with tempfile.TemporaryDirectory() as td:
                src = os.path.join(td, "scores.tsv")
                Path(src).write_text(text)
                with warnings.catch_warnings(record=True) as wl:
                    warnings.simplefilter("always")
                    outlier(score_file=src, output_prefix=os.path.join(td, "o"), quantile=quantile)
                files = {f[2:]: Path(td, f).read_text() for f in sorted(os.listdir(td)) if f.startswith("o.")}
                out.append(dict(table=name, quantile=quantile, files=files,
                                warnings=[str(w.message).replace(td, "") for w in wl if w.category is UserWarning]))
    (OUT / "outlier.json").write_text(json.dumps(dict(tables=tables, runs=out), separators=(",", ":")) + "\n")
    print("outlier.json", len(out))


if __name__ == "__main__" and os.path.isdir(REF):
    make_outlier()


# ---------------------------------------------------------------------------
# 7. ABBA-BABA family: fd / df / Danc / Dplus (sai/stats/*_statistic.py)
# ---------------------------------------------------------------------------

FOURPOP_SEEDED = [
    # name, seed, n_sites, n_ref, n_tgt, src_sizes, n_out, ploidies [ref, tgt, [src], out], missing rate
    ("small_out", 1, 40, 6, 5, [2], 2, [2, 2, [2], 2], 0.0),
    ("small_noout", 2, 40, 6, 5, [2], 0, [2, 2, [2], 2], 0.0),
    ("two_src_missing", 3, 200, 9, 7, [1, 3], 2, [2, 2, [2, 1], 2], 0.05),
    ("nan_sites", 4, 60, 3, 3, [1], 1, [2, 2, [2], 2], 0.5),
    ("mixed_ploidy", 5, 130, 4, 6, [2, 2], 3, [4, 1, [2, 3], 2], 0.01),
    ("blocks_129", 6, 129, 5, 5, [2], 2, [2, 2, [2], 2], 0.0),
    ("blocks_1000", 7, 1000, 5, 5, [2], 2, [2, 2, [2], 2], 0.0),
    ("beyond_buffer_9000", 8, 9000, 4, 4, [1], 1, [2, 2, [2], 2], 0.0),
    ("seven_sites", 9, 7, 5, 5, [2], 2, [2, 2, [2], 2], 0.0),
]


def make_fourpop():
    sys.path.insert(0, str(OUT))
    from seeded import fourpop_inputs
    from sai.stats import DancStatistic, DfStatistic, DplusStatistic, FdStatistic

    A = np.array
    classes = {"fd": FdStatistic, "df": DfStatistic, "Danc": DancStatistic, "Dplus": DplusStatistic}
    cases = []

    def run(name, ref, tgt, srcs, out, pl, inline):
        kw = dict(ref_gts=ref, tgt_gts=tgt, src_gts_list=srcs, out_gts=out, ref_ploidy=pl[0], tgt_ploidy=pl[1],
                  src_ploidy_list=pl[2], out_ploidy=pl[3])
        res = {k: [hx(v) for v in cls(**kw).compute()["value"]] for k, cls in classes.items()}
        rec = dict(name=name, ploidies=pl, out=res)
        if inline:
            rec.update(ref_gts=ints(ref), tgt_gts=ints(tgt), src_gts_list=[ints(s) for s in srcs],
                       out_gts=None if out is None else ints(out))
        cases.append(rec)
        return res

    # the reference tests' inputs and published answers
    r = run("fd_test", A([[0, 1], [1, 0], [0, 1]]), A([[1, 0], [0, 1], [1, 0]]), [A([[1, 1], [1, 1], [1, 1]])], None,
            [1, 1, [1], None], True)
    assert np.isclose(float.fromhex(r["fd"][0]), 0)  # tests/stats/test_fd_statistic.py
    ref, tgt, src = A([[0, 0], [0, 0], [1, 1]]), A([[1, 0], [0, 1], [0, 1]]), A([[0, 1], [1, 0], [1, 0]])
    r = run("df_danc_dplus_test", ref, tgt, [src], None, [1, 1, [1], None], True)
    assert np.isclose(float.fromhex(r["df"][0]), 0.2)  # test_df_statistic.py:71
    assert np.isclose(float.fromhex(r["Danc"][0]), -1 / 3)  # test_danc_statistic.py:48
    assert np.isclose(float.fromhex(r["Dplus"][0]), 0)  # test_dplus_statistic.py:77
    run("with_out_ploidy1", ref, tgt, [src], A([[0, 0], [1, 0], [0, 0]]), [1, 1, [1], 1], True)
    run("zero_denominators", A([[0, 0]]), A([[0, 0]]), [A([[0, 0]])], None, [1, 1, [1], None], True)
    for name, seed, n_sites, n_ref, n_tgt, src_sizes, n_out, pl, miss in FOURPOP_SEEDED:
        ref, tgt, srcs, out = fourpop_inputs(seed, n_sites, n_ref, n_tgt, src_sizes, n_out, pl, miss)
        cases.append(dict(run(name, ref, tgt, srcs, out, pl, False) and cases.pop(), seeded=[seed, n_sites, n_ref, n_tgt, src_sizes, n_out, miss]))
    (OUT / "fourpop_cases.json").write_text(json.dumps(cases, separators=(",", ":")) + "\n")
    print("fourpop_cases.json", len(cases))


def make_pipeline_outgroup():
    """WindowGenerator + FeaturePreprocessor + process_items with an outgroup, two sources and the
    four ABBA-BABA statistics next to U and Q (multi-source column expansion, header rule)."""
    sys.path.insert(0, str(OUT))
    from itertools import combinations

    from seeded import fourpop_inputs
    from sai.configs import PloidyConfig, StatConfig
    from sai.generators import WindowGenerator
    from sai.preprocessors import FeaturePreprocessor
    from sai.utils import split_genome
    from sai.utils.genomic_dataclasses import ChromosomeData

    out = []
    for name, seed, n_src, with_out in (("outgroup_two_src", 21, 2, True), ("no_outgroup_one_src", 22, 1, False)):
        n_sites = 700
        pl = [2, 2, [2] * n_src, 2]
        ref, tgt, srcs, og = fourpop_inputs(seed, n_sites, 8, 6, [1] * n_src, 2 if with_out else 0, pl, 0.02)
        rng = np.random.default_rng(seed + 100)
        pos = (np.cumsum(rng.integers(1, 40, size=n_sites)) + 500).astype(np.int32)
        ploidies = {"ref": {"R": 2}, "tgt": {"T": 2}, "src": {f"S{i}": 2 for i in range(n_src)}}
        if with_out:
            ploidies["outgroup"] = {"O": 2}
        stats = {"fd": True, "DD": not with_out, "U": {"ref": {"R": 0.4}, "tgt": {"T": 0.3}, "src": {f"S{i}": ">=0.5" for i in range(n_src)}},
                 "df": True, "Danc": True, "Q": {"ref": {"R": 0.4}, "tgt": {"T": 0.9}, "src": {f"S{i}": ">=0.5" for i in range(n_src)}},
                 "Dplus": True}
        wg = object.__new__(WindowGenerator)
        wg.win_len, wg.win_step, wg.chr_name, wg.ploidy_config = 3000, 1500, "9", PloidyConfig(ploidies)
        mk = lambda g: ChromosomeData(POS=pos.copy(), REF=None, ALT=None, GT=g.copy())
        wg.ref_data, wg.ref_samples = {"R": mk(ref)}, {"R": []}
        wg.tgt_data, wg.tgt_samples = {"T": mk(tgt)}, {"T": []}
        wg.src_data = {f"S{i}": mk(s) for i, s in enumerate(srcs)}
        wg.src_samples = {f"S{i}": [] for i in range(n_src)}
        wg.out_data, wg.out_samples = ({"O": mk(og)}, {"O": []}) if with_out else (None, None)
        wg.num_src = n_src
        wg.src_combinations = list(combinations(wg.src_samples.keys(), n_src))
        wg.tgt_windows = {"T": split_genome(pos=pos, window_size=3000, step_size=1500)}
        sc = StatConfig(json.loads(json.dumps(stats)))
        with tempfile.TemporaryDirectory() as td:
            tsv = os.path.join(td, "o.tsv")
            fp = FeaturePreprocessor(output_file=tsv, stat_config=sc, anc_allele_available=True)
            items = []
            for item in wg.get():
                items.extend(fp.run(**item))
            fp.process_items(items)
            text = {"tsv": open(tsv).read(), "U": open(os.path.join(td, "o.U.log")).read(),
                    "Q": open(os.path.join(td, "o.Q.log")).read()}
        out.append(dict(name=name, seed=seed, n_src=n_src, with_out=with_out, n_sites=n_sites, ploidies=ploidies,
                        stats=stats, pos=ints(pos), n_windows=len(items), text=text,
                        items=[{k: ([hx(v) for v in it[k]] if isinstance(it[k], list) else hx(it[k]))
                                for k in ("fd", "df", "Danc", "Dplus", "DD") if k in it}
                               | {"out_pop": it["out_pop"], "nsnps": it["nsnps"]} for it in items]))
        print(name, len(items), "windows")
    (OUT / "pipeline_outgroup.json").write_text(json.dumps(out, separators=(",", ":")) + "\n")


if __name__ == "__main__" and os.path.isdir(REF):
    make_fourpop()
    make_pipeline_outgroup()


# ---------------------------------------------------------------------------
# 8. DD (sai/stats/dd_statistic.py; needs scipy, which this image has)
# ---------------------------------------------------------------------------

DD_SEEDED = [
    # name, seed, n_sites, n_ref, n_tgt, src_sizes, missing rate, ploidies
    ("dd_small", 31, 50, 6, 5, [2], 0.0, [2, 2, [2], 2]),
    ("dd_missing", 32, 300, 9, 7, [1, 3], 0.1, [2, 2, [2, 2], 2]),
    ("dd_many_src_individuals", 33, 120, 5, 4, [11, 150], 0.02, [2, 2, [2, 1], 2]),
    ("dd_tetraploid", 34, 80, 3, 8, [2], 0.05, [4, 4, [4], 2]),
    ("dd_long", 35, 5000, 4, 3, [2], 0.01, [2, 2, [2], 2]),
]


def make_dd():
    sys.path.insert(0, str(OUT))
    from seeded import fourpop_inputs
    from sai.stats import DdStatistic

    A = np.array
    cases = []
    r = DdStatistic(ref_gts=A([[1, 1], [0, 0]]), tgt_gts=A([[1, 0], [0, 1]]), src_gts_list=[A([[0, 1], [1, 1]])],
                    ref_ploidy=1, tgt_ploidy=1, src_ploidy_list=[1]).compute()
    assert r["name"] == "DD" and np.isclose(r["value"][0], 0.5)  # tests/stats/test_dd_statistic.py:40
    cases.append(dict(name="dd_test", ref_gts=[[1, 1], [0, 0]], tgt_gts=[[1, 0], [0, 1]], src_gts_list=[[[0, 1], [1, 1]]],
                      out=[hx(v) for v in r["value"]]))
    for name, seed, n_sites, n_ref, n_tgt, src_sizes, miss, pl in DD_SEEDED:
        ref, tgt, srcs, _ = fourpop_inputs(seed, n_sites, n_ref, n_tgt, src_sizes, 0, pl, miss)
        r = DdStatistic(ref_gts=ref, tgt_gts=tgt, src_gts_list=srcs, ref_ploidy=pl[0], tgt_ploidy=pl[1],
                        src_ploidy_list=pl[2]).compute()
        cases.append(dict(name=name, seeded=[seed, n_sites, n_ref, n_tgt, src_sizes, miss], ploidies=pl,
                          out=[hx(v) for v in r["value"]]))
    (OUT / "dd_cases.json").write_text(json.dumps(cases, separators=(",", ":")) + "\n")
    print("dd_cases.json", len(cases))


if __name__ == "__main__" and os.path.isdir(REF):
    make_dd()


# ---------------------------------------------------------------------------
# 9. random chromosomes / configs through the reference's own WindowGenerator +
#    FeaturePreprocessor + process_items (several ref / tgt populations, two
#    sources, outgroups, ploidy 1-4, chunk bounds, all seven statistics); only
#    the seeds and the output text are stored, seeded.fuzz_scenario rebuilds
#    the inputs
# ---------------------------------------------------------------------------

FUZZ_SEEDS = sorted({*range(100, 132), 300})


def make_pipeline_fuzz():
    sys.path.insert(0, str(OUT))
    from itertools import combinations

    from seeded import fuzz_scenario
    from sai.configs import PloidyConfig, StatConfig
    from sai.generators import WindowGenerator
    from sai.preprocessors import FeaturePreprocessor
    from sai.utils import split_genome
    from sai.utils.genomic_dataclasses import ChromosomeData

    out = []
    for seed in FUZZ_SEEDS:
        sc = fuzz_scenario(seed)
        pos, start, end = sc["pos"], sc["start"], sc["end"]
        sel = np.ones(len(pos), bool) if start is None else (pos >= start) & (pos <= end)
        if not sel.any():
            continue
        wg = object.__new__(WindowGenerator)
        wg.win_len, wg.win_step, wg.chr_name, wg.ploidy_config = sc["win"], sc["step"], "7", PloidyConfig(sc["pl"])
        for g in ("ref", "tgt", "src"):
            setattr(wg, f"{g}_data", {k: ChromosomeData(POS=pos[sel].copy(), REF=None, ALT=None, GT=v[sel].copy())
                                      for k, v in sc["gts"][g].items()})
            setattr(wg, f"{g}_samples", {k: [] for k in sc["gts"][g]})
        if sc["gts"]["outgroup"]:
            wg.out_data = {k: ChromosomeData(POS=pos[sel].copy(), REF=None, ALT=None, GT=v[sel].copy())
                           for k, v in sc["gts"]["outgroup"].items()}
            wg.out_samples = {k: [] for k in sc["gts"]["outgroup"]}
        else:
            wg.out_data = wg.out_samples = None
        wg.num_src = len(sc["gts"]["src"])
        wg.src_combinations = list(combinations(wg.src_samples.keys(), wg.num_src))
        wg.tgt_windows = {
            t: split_genome(pos=(wg.tgt_data[t].POS if start is None and end is None
                                 else [start, end - sc["win"] + sc["step"]]),
                            window_size=sc["win"], step_size=sc["step"], start=start)
            for t in wg.tgt_samples
        }
        stat_config = StatConfig(json.loads(json.dumps(sc["stats"])))
        with tempfile.TemporaryDirectory() as td:
            tsv = os.path.join(td, "o.tsv")
            fp = FeaturePreprocessor(output_file=tsv, stat_config=stat_config, anc_allele_available=sc["anc"])
            items = []
            for item in wg.get():
                items.extend(fp.run(**item))
            fp.process_items(items)
            text = {"tsv": open(tsv).read()}
            for k in ("U", "Q"):
                text[k] = open(os.path.join(td, f"o.{k}.log")).read()
        out.append(dict(seed=seed, n_items=len(items), text=text))
        print("fuzz", seed, len(items), "items", list(sc["stats"]))
    (OUT / "pipeline_fuzz.json").write_text(json.dumps(out, separators=(",", ":")) + "\n")


if __name__ == "__main__" and os.path.isdir(REF):
    make_pipeline_fuzz()


# ---------------------------------------------------------------------------
# 10. populations with different site sets / repeated positions through the
#     reference's WindowGenerator._window_generator + FeaturePreprocessor
#     (window_generator.py:193-231): output text, or the exception the
#     reference raises (type and message -- the message depends on the numpy
#     version, the tests compare the type)
# ---------------------------------------------------------------------------

SITESET_CASES = [("ragged", 1), ("ragged", 2), ("ragged", 3), ("dup_u", 4), ("dup_u", 5), ("dup_u", 6), ("dup_uq", 7),
                 ("dup_uq", 8), ("dup_uneven", 9), *[("dup_rare", s) for s in range(10, 18)],
                 ("unsorted", 18), ("unsorted", 19), ("unsorted", 20), ("unsorted_mixed", 21), ("unsorted_mixed", 22),
                 ("unsorted_mixed", 23)]


def make_sitesets():
    sys.path.insert(0, str(OUT))
    from itertools import combinations

    from seeded import siteset_scenario
    from sai.configs import PloidyConfig, StatConfig
    from sai.generators import WindowGenerator
    from sai.preprocessors import FeaturePreprocessor
    from sai.utils import split_genome
    from sai.utils.genomic_dataclasses import ChromosomeData

    out = []
    for kind, seed in SITESET_CASES:
        sc = siteset_scenario(kind, seed)
        wg = object.__new__(WindowGenerator)
        wg.win_len, wg.win_step, wg.chr_name, wg.ploidy_config = sc["win"], sc["step"], "5", PloidyConfig(sc["pl"])
        for g in ("ref", "tgt", "src"):
            setattr(wg, f"{g}_data", {k: ChromosomeData(POS=sc["pos"][g][k].copy(), REF=None, ALT=None, GT=v.copy())
                                      for k, v in sc["gts"][g].items()})
            setattr(wg, f"{g}_samples", {k: [] for k in sc["gts"][g]})
        wg.out_data = wg.out_samples = None
        wg.num_src = len(sc["gts"]["src"])
        wg.src_combinations = list(combinations(wg.src_samples.keys(), wg.num_src))
        wg.tgt_windows = {t: split_genome(pos=wg.tgt_data[t].POS, window_size=sc["win"], step_size=sc["step"])
                          for t in wg.tgt_samples}
        stat_config = StatConfig(json.loads(json.dumps(sc["stats"])))
        rec = dict(kind=kind, seed=seed)
        with tempfile.TemporaryDirectory() as td:
            tsv = os.path.join(td, "o.tsv")
            fp = FeaturePreprocessor(output_file=tsv, stat_config=stat_config, anc_allele_available=sc["anc"])
            items = []
            try:
                for item in wg.get():
                    items.extend(fp.run(**item))
                fp.process_items(items)
                rec["text"] = {"tsv": open(tsv).read()}
                for k in ("U", "Q"):
                    f = os.path.join(td, f"o.{k}.log")
                    if os.path.exists(f):
                        rec["text"][k] = open(f).read()
                rec["n_items"] = len(items)
            except Exception as e:  # noqa: BLE001 -- the capture IS the reference's behaviour
                rec["error"] = [type(e).__name__, str(e)]
                rec["items_before_error"] = len(items)
        out.append(rec)
        print("siteset", kind, seed, rec.get("n_items"), rec.get("error"))
    (OUT / "sitesets.json").write_text(json.dumps(out, separators=(",", ":")) + "\n")


if __name__ == "__main__" and os.path.isdir(REF):
    make_sitesets()


# ---------------------------------------------------------------------------
# 11. more than six source populations (stat_utils.py:114-119, 141-152 loop over any number): seeded
#     chromosomes with seven to ten sources through the reference's WindowGenerator + FeaturePreprocessor
#     (output text), and single windows through UStatistic / QStatistic
# ---------------------------------------------------------------------------

MANY_SOURCE_SEEDS = [900, 901, 902, 903, 904, 905]


def make_many_sources():
    sys.path.insert(0, str(OUT))
    from itertools import combinations

    from seeded import many_sources_scenario
    from sai.configs import PloidyConfig, StatConfig
    from sai.generators import WindowGenerator
    from sai.preprocessors import FeaturePreprocessor
    from sai.stats import QStatistic, UStatistic
    from sai.utils import split_genome
    from sai.utils.genomic_dataclasses import ChromosomeData

    out = []
    for seed in MANY_SOURCE_SEEDS:
        sc = many_sources_scenario(seed)
        pos = sc["pos"]
        wg = object.__new__(WindowGenerator)
        wg.win_len, wg.win_step, wg.chr_name, wg.ploidy_config = sc["win"], sc["step"], "7", PloidyConfig(sc["pl"])
        for g in ("ref", "tgt", "src"):
            setattr(wg, f"{g}_data", {k: ChromosomeData(POS=pos.copy(), REF=None, ALT=None, GT=v.copy()) for k, v in sc["gts"][g].items()})
            setattr(wg, f"{g}_samples", {k: [] for k in sc["gts"][g]})
        if sc["gts"]["outgroup"]:
            wg.out_data = {k: ChromosomeData(POS=pos.copy(), REF=None, ALT=None, GT=v.copy()) for k, v in sc["gts"]["outgroup"].items()}
            wg.out_samples = {k: [] for k in sc["gts"]["outgroup"]}
        else:
            wg.out_data = wg.out_samples = None
        wg.num_src = len(sc["gts"]["src"])
        wg.src_combinations = list(combinations(wg.src_samples.keys(), wg.num_src))
        wg.tgt_windows = {t: split_genome(pos=wg.tgt_data[t].POS, window_size=sc["win"], step_size=sc["step"], start=None)
                          for t in wg.tgt_samples}
        stat_config = StatConfig(json.loads(json.dumps(sc["stats"])))
        with tempfile.TemporaryDirectory() as td:
            tsv = os.path.join(td, "o.tsv")
            fp = FeaturePreprocessor(output_file=tsv, stat_config=stat_config, anc_allele_available=sc["anc"])
            items = []
            for item in wg.get():
                items.extend(fp.run(**item))
            fp.process_items(items)
            text = {"tsv": open(tsv).read()}
            for k in ("U", "Q"):
                text[k] = open(os.path.join(td, f"o.{k}.log")).read()
        # the whole chromosome as ONE window through the statistic classes (the per-window plugin route)
        t0 = next(iter(sc["gts"]["tgt"]))
        kw = dict(ref_gts=sc["gts"]["ref"]["R0"], tgt_gts=sc["gts"]["tgt"][t0], src_gts_list=list(sc["gts"]["src"].values()),
                  ref_ploidy=2, tgt_ploidy=sc["pl"]["tgt"][t0], src_ploidy_list=list(sc["pl"]["src"].values()))
        up = stat_config.get_parameters("U")
        qp = stat_config.get_parameters("Q")
        u = UStatistic(**kw).compute(pos=pos, w=up["ref"]["R0"], x=up["tgt"][t0], y_list=list(up["src"].values()),
                                     anc_allele_available=sc["anc"])
        q = QStatistic(**kw).compute(pos=pos, w=qp["ref"]["R0"], quantile=qp["tgt"][t0], y_list=list(qp["src"].values()),
                                     anc_allele_available=sc["anc"])
        assert any(it["U"] > 0 for it in items if isinstance(it["U"], int)), seed  # some window passes all the sources' conditions
        out.append(dict(seed=seed, n_src=wg.num_src, n_items=len(items), text=text,
                        whole=dict(tgt=t0, U=u["value"], U_cdd_pos=ints(u["cdd_pos"]), Q=hx(q["value"]), Q_cdd_pos=ints(q["cdd_pos"]))))
        print("many sources", seed, wg.num_src, "sources", len(items), "items", list(sc["stats"]))
    (OUT / "many_sources.json").write_text(json.dumps(out, separators=(",", ":")) + "\n")


if __name__ == "__main__" and os.path.isdir(REF):
    make_many_sources()
