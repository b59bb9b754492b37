"""GPU parity tests of the HIP kernels, called through the C ABI (sai_amd._ffi / Engine),
against the numpy oracle and the committed golden vectors.  Bit-exact everywhere: integer
counts, candidate positions, and the f64 frequencies / Q values."""

import os

import numpy as np
import pytest

from conftest import load_golden, same_f64, stat_case_inputs, unhex

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from sai_amd.engine import Engine

    return Engine.get(0)


def tile_numpy(g8: np.ndarray) -> np.ndarray:
    n_sites, n_ind = g8.shape
    n_tiles = (n_sites + 63) // 64
    pad = np.zeros((n_tiles * 64, n_ind), dtype=np.int8)
    pad[:n_sites] = g8
    return pad.reshape(n_tiles, 64, n_ind).transpose(0, 2, 1).reshape(-1).copy()


def counts_numpy(g):
    present = g >= 0
    return np.where(present, g, 0).sum(axis=1).astype(np.int64), present.sum(axis=1).astype(np.int64)


@pytest.mark.parametrize("shape", [(1, 1), (63, 5), (64, 16), (65, 17), (200, 64), (130, 129), (1000, 3)])
def test_tile_layout(eng, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    g = rng.integers(-3, 5, size=shape).astype(np.int8)
    t = eng.tile(g)
    assert np.array_equal(t.tiles.cpu().numpy(), tile_numpy(g))
    # a wider integer input is narrowed on the host, values kept
    t2 = eng.tile(g.astype(np.int64))
    assert np.array_equal(t2.tiles.cpu().numpy(), tile_numpy(g))


@pytest.mark.parametrize("n_ind", [1, 2, 15, 16, 17, 31, 200, 1000, 1001, 4100])
def test_site_counts_exact(eng, n_ind):
    rng = np.random.default_rng(n_ind)
    n_sites = 333
    g = rng.integers(0, 3, size=(n_sites, n_ind)).astype(np.int8)
    g[rng.random(g.shape) < 0.05] = -2
    g[rng.random(g.shape) < 0.01] = -128
    g[rng.random(g.shape) < 0.01] = 127
    g[5] = 127  # worst case for the 16-bit accumulators
    g[6] = -1  # everything missing
    other = rng.integers(-1, 2, size=(n_sites, 7)).astype(np.int8)
    counts = eng.site_counts([eng.tile(g), eng.tile(other), eng.tile(g)]).cpu().numpy().astype(np.int64)
    for p, m in enumerate([g, other, g]):
        s, c = counts_numpy(m.astype(np.int64))
        assert np.array_equal(counts[p, :, 0], s)
        assert np.array_equal(counts[p, :, 1], c)


@pytest.mark.parametrize("top,rate", [(2, 0.0), (63, 0.0), (64, 0.0), (2, 2e-4), (63, 2e-4), (127, 2e-3)])
def test_site_counts_groups_with_and_without_missing_calls(eng, top, rate):
    """The stream loop adds four rows bytewise when a group holds no missing call and no dosage of 64 or more, and goes
    call by call otherwise (stream_loops.hpp acc_group): dosages up to the limit of either way, missing calls and large
    dosages rare enough that both kinds of group meet in one tile, full and partial groups, one load and many."""
    rng = np.random.default_rng(top * 1000 + int(rate * 1e6))
    n_sites = 700
    for n_ind in (3, 16, 50, 64, 67, 200, 1000):
        g = rng.integers(0, top + 1, size=(n_sites, n_ind)).astype(np.int8)
        g[0] = top  # every call at the top of the range: the bytewise sums at their limit
        if rate:
            hit = rng.random(g.shape) < rate
            g[hit] = rng.choice(np.array([-1, -128, -2, 127, 64, 100], dtype=np.int8), size=int(hit.sum()))
        counts = eng.site_counts([eng.tile(g)]).cpu().numpy().astype(np.int64)
        s, c = counts_numpy(g.astype(np.int64))
        assert np.array_equal(counts[0, :, 0], s) and np.array_equal(counts[0, :, 1], c), (n_ind, top, rate)


STATS = load_golden("stats_cases.json")


@pytest.mark.parametrize("case", STATS, ids=[c["name"] for c in STATS])
def test_golden_stats_through_plugin_api(eng, case):
    from sai_amd.stats import QStatistic, UStatistic, calc_freq, compute_matching_loci

    a = stat_case_inputs(case)
    out = case["out"]
    mats = [a["ref_gts"], a["tgt_gts"]] + a["src_gts_list"]
    for m, p, exp in zip(mats, a["ploidy"], out["raw_freq"]):
        got = calc_freq(m, p)
        assert all(same_f64(g, unhex(e)) for g, e in zip(got, exp))
    rf, tf, cond = compute_matching_loci(
        a["ref_gts"], a["tgt_gts"], a["src_gts_list"], a["w"], a["y_list"], a["ploidy"], a["anc"]
    )
    assert all(same_f64(g, unhex(e)) for g, e in zip(rf, out["ref_freq"]))
    assert all(same_f64(g, unhex(e)) for g, e in zip(tf, out["tgt_freq"]))
    assert cond.tolist() == out["condition"]
    kw = dict(
        ref_gts=a["ref_gts"], tgt_gts=a["tgt_gts"], src_gts_list=a["src_gts_list"], ref_ploidy=a["ploidy"][0],
        tgt_ploidy=a["ploidy"][1], src_ploidy_list=a["ploidy"][2:],
    )  # fmt: skip
    u = UStatistic(**kw).compute(pos=a["pos"], w=a["w"], x=a["x"], y_list=a["y_list"], anc_allele_available=a["anc"])
    assert u["name"] == "U" and isinstance(u["value"], int) and u["value"] == out["U"]
    assert u["cdd_pos"].tolist() == out["U_cdd_pos"]
    q = QStatistic(**kw).compute(
        pos=a["pos"], w=a["w"], y_list=a["y_list"], quantile=a["quantile"], anc_allele_available=a["anc"]
    )
    assert q["name"] == "Q" and same_f64(q["value"], unhex(out["Q"]))
    assert np.asarray(q["cdd_pos"]).astype(np.int64).tolist() == out["Q_cdd_pos"]
    assert str(q["cdd_pos"].dtype) == out["Q_cdd_dtype"]


def test_plugin_errors_match_reference(eng):
    from sai_amd.stats import QStatistic, UStatistic, calc_freq, compute_matching_loci

    g = load_golden("errors_and_freq.json")
    A = np.array
    ref, tgt = A([[0, 1, 0], [1, 1, 0], [0, 0, 1]]), A([[1, 1, 0], [0, 1, 1], [1, 1, 1]])
    srcs = [A([[0, 0, 1], [1, 1, 0], [0, 1, 1]]), A([[1, 1, 0], [1, 0, 0], [1, 1, 0]])]
    y2 = [("=", 0.5), ("=", 0.5)]
    kw = dict(ref_gts=ref, tgt_gts=tgt, src_gts_list=srcs[:1], ref_ploidy=3, tgt_ploidy=1, src_ploidy_list=[2])
    calls = {
        "w_low": lambda: compute_matching_loci(ref, tgt, srcs, -0.1, y2, [2, 2, 2], False),
        "w_high": lambda: compute_matching_loci(ref, tgt, srcs, 1.1, y2, [2, 2, 2], False),
        "y_low": lambda: compute_matching_loci(ref, tgt, srcs, 0.5, [("=", -0.1)], [2, 2, 2], False),
        "y_high": lambda: compute_matching_loci(ref, tgt, srcs, 0.5, [("=", 1.1)], [2, 2, 2], False),
        "bad_op": lambda: compute_matching_loci(ref, tgt, srcs, 0.5, [("invalid", 0.5)], [2, 2, 2], False),
        "len_mismatch": lambda: compute_matching_loci(ref, tgt, srcs, 0.5, [("=", 0.5)], [2, 2, 2], False),
        "ploidy_none": lambda: calc_freq(ref, ploidy=None),
        "ploidy_float": lambda: calc_freq(ref, ploidy=9.9),
        "ploidy_neg": lambda: calc_freq(ref, ploidy=-100),
        "u_missing_kw": lambda: UStatistic(**kw).compute(pos=A([0, 1, 2]), w=0.5, x=0.5, y_list=[("=", 0)]),
        "q_missing_kw": lambda: QStatistic(**kw).compute(pos=A([0, 1, 2]), w=0.5, quantile=0.95, anc_allele_available=False),
    }  # fmt: skip
    for rec in g["errors"]:
        with pytest.raises(ValueError) as ei:
            calls[rec["label"]]()
        assert str(ei.value) == rec["msg"]


def _window_pass(eng, mats, ploidy, sets, pos, starts, ends):
    import torch

    pops = [eng.tile(m) for m in mats]
    counts = eng.site_counts(pops)
    tgt_freq, flags, _ = eng.site_flags(counts, ploidy, sets)
    pos_dev = torch.as_tensor(pos.astype(np.int32)).to(eng.device)
    lo, hi = eng.window_bounds(pos_dev, starts, ends)
    return eng.window_stats(tgt_freq, flags, sets, lo, hi, pos=pos_dev, cap_hint=8), lo.cpu().numpy(), hi.cpu().numpy()


def test_batched_windows_vs_oracle(eng):
    """Many overlapping windows, two parameter sets, missing data: records and candidate lists
    equal the per-window oracle (the reference's structure) bit for bit."""
    from oracle import sai_oracle as O
    from sai_amd import _ffi

    rng = np.random.default_rng(99)
    n_sites = 6000
    pos = np.cumsum(rng.integers(1, 50, n_sites)).astype(np.int64)
    p = rng.random(n_sites) ** 4
    intro = rng.random(n_sites) < 0.02
    pr, pt = p.copy(), p.copy()
    pr[intro] = 0.0
    pt[intro] = 0.2 + 0.7 * rng.random(int(intro.sum()))
    ref = rng.binomial(2, pr[:, None], size=(n_sites, 50)).astype(np.int64)
    tgt = rng.binomial(2, pt[:, None], size=(n_sites, 37)).astype(np.int64)
    src = rng.binomial(2, p[:, None], size=(n_sites, 2)).astype(np.int64)
    src[intro] = 2
    for m in (ref, tgt, src):
        m[rng.random(m.shape) < 0.01] = -2
    windows = O.split_windows(pos, 5000, 1000)
    starts = np.array([w[0] for w in windows])
    ends = np.array([w[1] for w in windows])
    specs = [
        dict(w=0.05, x=0.3, quantile=0.95, y_list=[("=", 1.0)], anc=True),
        dict(w=0.5, x=0.1, quantile=0.5, y_list=[(">=", 0.5)], anc=False),
        dict(w=1.0, x=0.0, quantile=1.0, y_list=[("<=", 1.0)], anc=True),  # nearly every site qualifies
    ]
    sets = [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in specs]
    res, lo, hi = _window_pass(eng, [ref, tgt, src], [2, 2, 2], sets, pos, starts, ends)
    assert res.records.shape == (3, len(windows))
    for si, s in enumerate(specs):
        for wi, (ws, we) in enumerate(windows):
            m = (pos >= ws) & (pos <= we)
            assert (lo[wi], hi[wi]) == (np.searchsorted(pos, ws), np.searchsorted(pos, we, side="right"))
            kw = dict(ref_gts=ref[m], tgt_gts=tgt[m], src_gts_list=[src[m]], ref_ploidy=2, tgt_ploidy=2,
                      src_ploidy_list=[2], pos=pos[m], w=s["w"], y_list=s["y_list"], anc_allele_available=s["anc"])  # fmt: skip
            eu = O.u_stat(x=s["x"], **kw)
            eq = O.q_stat(quantile=s["quantile"], **kw)
            rec = res.records[si, wi]
            assert rec["n_sites"] == int(m.sum())
            assert rec["u_count"] == eu["value"]
            assert res.u_list(si, wi).tolist() == eu["cdd_pos"].tolist()
            assert same_f64(rec["q"], eq["value"]), (si, wi, rec, eq["value"])
            assert res.q_list(si, wi).tolist() == np.asarray(eq["cdd_pos"]).astype(np.int64).tolist()


def test_quantile_beyond_lds_capacity(eng):
    """All selection paths of window_stats: <= 256 qualifying sites (wave kernel, rank counting),
    <= 4096 (workgroup kernel, radix select in LDS) and more (radix select re-reading the
    per-site arrays); always numpy's 'linear' quantile bit for bit."""
    from oracle import sai_oracle as O
    from sai_amd import _ffi

    rng = np.random.default_rng(3)
    n_sites = 20000
    ref = np.zeros((n_sites, 4), dtype=np.int64)
    tgt = rng.integers(0, 3, size=(n_sites, 23)).astype(np.int64)
    tgt[rng.random(tgt.shape) < 0.05] = -1
    src = np.full((n_sites, 1), 2, dtype=np.int64)
    pos = np.arange(1, n_sites + 1, dtype=np.int64) * 3
    for quantile in (0.0, 0.3, 0.95, 1.0):
        sets = [_ffi.make_params(0.5, 0.5, quantile, [("=", 1.0)], True)]
        ends = [10**9, 20000, 3000, 600, 768, 771, 3]
        res, _, _ = _window_pass(eng, [ref, tgt, src], [2, 2, 2], sets, pos, np.ones(len(ends), dtype=np.int64), np.array(ends))
        assert res.records[0]["n_cond"].tolist() == [20000, 6666, 1000, 200, 256, 257, 1]
        for wi, we in enumerate(ends):
            m = pos <= we
            kw = dict(ref_gts=ref[m], tgt_gts=tgt[m], src_gts_list=[src[m]], ref_ploidy=2, tgt_ploidy=2,
                      src_ploidy_list=[2], pos=pos[m], w=0.5, y_list=[("=", 1.0)], anc_allele_available=True)  # fmt: skip
            eq = O.q_stat(quantile=quantile, **kw)
            eu = O.u_stat(x=0.5, **kw)
            rec = res.records[0, wi]
            assert rec["n_cond"] == int(m.sum()) and rec["u_count"] == eu["value"]
            assert same_f64(rec["q"], eq["value"])
            assert res.q_list(0, wi).tolist() == eq["cdd_pos"].tolist()
            assert res.u_list(0, wi).tolist() == eu["cdd_pos"].tolist()


def test_windows_of_a_million_sites(eng):
    """Windows far beyond anything a wave keeps: 1.2 million sites in one window, a third of them qualifying,
    47 distinct frequencies (tens of thousands of ties per value, so the digit histogram's bins are huge and
    the level-by-level select runs to the bottom), next to a window of three sites and an empty one; wave
    form (one set) and workgroup form (five sets)."""
    from oracle import sai_oracle as O
    from sai_amd import _ffi

    rng = np.random.default_rng(12)
    n_sites = 1_200_000
    ref = (rng.random((n_sites, 3)) < 0.15).astype(np.int64)
    tgt = rng.integers(0, 3, size=(n_sites, 23)).astype(np.int64)
    tgt[rng.random(tgt.shape) < 0.02] = -2
    src = np.full((n_sites, 1), 2, dtype=np.int64)
    src[rng.random(n_sites) < 0.3] = 0
    pos = np.cumsum(rng.integers(1, 4, n_sites)).astype(np.int64)
    starts = np.array([1, int(pos[500_000]), int(pos[7]), int(pos[-1]) + 5])
    ends = np.array([int(pos[-1]), int(pos[900_000]), int(pos[9]), int(pos[-1]) + 50])
    specs = [dict(w=0.5, x=0.5, quantile=q, y_list=[("=", 1.0)], anc=True) for q in (0.95, 0.0, 0.5, 1.0, 0.123456)]
    for group in (specs[:1], specs):
        sets = [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in group]
        res, lo, hi = _window_pass(eng, [ref, tgt, src], [1, 2, 2], sets, pos, starts, ends)
        assert (hi - lo).tolist() == [n_sites, 400_001, 3, 0]
        for si, s in enumerate(group):
            for wi in range(3):
                m = (pos >= starts[wi]) & (pos <= ends[wi])
                kw = dict(ref_gts=ref[m], tgt_gts=tgt[m], src_gts_list=[src[m]], ref_ploidy=1, tgt_ploidy=2, src_ploidy_list=[2],
                          pos=pos[m], w=s["w"], y_list=s["y_list"], anc_allele_available=True)  # fmt: skip
                eq, eu = O.q_stat(quantile=s["quantile"], **kw), O.u_stat(x=s["x"], **kw)
                rec = res.records[si, wi]
                assert rec["u_count"] == eu["value"] and same_f64(rec["q"], eq["value"]), (si, wi, rec["q"], eq["value"])
                assert np.array_equal(res.u_list(si, wi), eu["cdd_pos"]) and np.array_equal(res.q_list(si, wi), eq["cdd_pos"])
            assert res.records[si, 3]["n_sites"] == 0 and np.isnan(res.records[si, 3]["q"])
        assert res.records[0, 0]["n_cond"] > 300_000


def test_shared_form_beyond_its_lds_capacities(eng):
    """window_stats / window_lists with >= 4 sets (one workgroup per window, the window's rows and stored
    frequencies in LDS): windows that fit, windows whose stored frequencies exceed the 1 024 kept in LDS
    (rows in LDS, frequencies read where they lie), windows of more than 128 tiles or 1 024 row words (all
    from global memory), light and heavy sets side by side, with and without inverted planes -- every
    record and both lists equal the per-window oracle."""
    from oracle import sai_oracle as O
    from sai_amd import _ffi

    rng = np.random.default_rng(41)
    n_sites = 12000
    pos = np.cumsum(rng.integers(1, 9, n_sites)).astype(np.int64)
    p = rng.random(n_sites) ** 2
    ref = rng.binomial(2, (p * 0.05)[:, None], size=(n_sites, 20)).astype(np.int64)
    tgt = rng.binomial(2, p[:, None], size=(n_sites, 31)).astype(np.int64)
    tgt[rng.random(tgt.shape) < 0.03] = -2
    s1 = rng.integers(0, 3, size=(n_sites, 1)).astype(np.int64)
    s2 = rng.integers(0, 3, size=(n_sites, 1)).astype(np.int64)
    ends = [int(pos[-1]), int(pos[9000]), int(pos[5000]), int(pos[1500]), int(pos[700]), int(pos[63]), int(pos[5])]
    starts = [1, int(pos[100]), 1, int(pos[300]), 1, 1, 1]
    for anc in (True, False):
        specs = [dict(w=1.0, x=0.4, quantile=0.95, y_list=[(">=", 0.0), (">=", 0.0)], anc=anc),  # nearly every site
                 dict(w=0.2, x=0.5, quantile=0.5, y_list=[("=", 1.0), (">=", 0.5)], anc=anc),
                 dict(w=0.05, x=0.1, quantile=0.0, y_list=[("=", 0.0), ("=", 0.0)], anc=anc),
                 dict(w=1.0, x=0.9, quantile=1.0, y_list=[("<=", 0.5), (">", 0.0)], anc=anc),
                 dict(w=0.5, x=0.3, quantile=0.777, y_list=[(">=", 0.5), ("<", 1.0)], anc=anc)]
        sets = [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in specs]
        res, lo, hi = _window_pass(eng, [ref, tgt, s1, s2], [2, 2, 2, 2], sets, pos, np.array(starts), np.array(ends))
        assert (hi - lo).max() > 128 * 64 and res.records[0]["n_cond"].max() > 1024 and res.records[0]["n_cond"].min() <= 6
        for si, s in enumerate(specs):
            for wi, (ws, we) in enumerate(zip(starts, ends)):
                m = (pos >= ws) & (pos <= we)
                kw = dict(ref_gts=ref[m], tgt_gts=tgt[m], src_gts_list=[s1[m], s2[m]], ref_ploidy=2, tgt_ploidy=2,
                          src_ploidy_list=[2, 2], pos=pos[m], w=s["w"], y_list=s["y_list"], anc_allele_available=anc)  # fmt: skip
                eq, eu = O.q_stat(quantile=s["quantile"], **kw), O.u_stat(x=s["x"], **kw)
                rec = res.records[si, wi]
                assert rec["n_sites"] == int(m.sum()) and rec["u_count"] == eu["value"], (anc, si, wi)
                assert same_f64(rec["q"], eq["value"]), (anc, si, wi, rec["q"], eq["value"])
                assert res.q_list(si, wi).tolist() == eq["cdd_pos"].tolist() and res.u_list(si, wi).tolist() == eu["cdd_pos"].tolist()


def test_synth_device_equals_host(eng):
    import ctypes as C

    from sai_amd import _ffi

    lib = _ffi.load()
    seed, chrom = 20260633, 3
    for pop_stream, n_ind, ploidy, mpm in [(0, 37, 2, 0), (1, 64, 2, 1000), (2, 2, 2, 1000), (3, 5, 4, 50000), (1, 9, 1, 0)]:
        n_sites, site0 = 1000, 12345
        host = np.empty((n_sites, n_ind), dtype=np.int8)
        _ffi.check(lib.sai_synth_fill_host(seed, chrom, site0, n_sites, pop_stream, n_ind, ploidy, mpm,
                                           host.ctypes.data_as(C.c_void_p)))  # fmt: skip
        dev = eng.synth_population(seed, chrom, site0, n_sites, pop_stream, n_ind, ploidy, mpm)
        assert np.array_equal(dev.tiles.cpu().numpy(), tile_numpy(host))
        assert host.max() <= ploidy and host.min() >= -ploidy
    gaps = np.empty(5000, dtype=np.int32)
    _ffi.check(lib.sai_synth_gaps_host(seed, chrom, 0, 5000, gaps.ctypes.data_as(C.c_void_p)))
    pos = eng.synth_positions(seed, chrom, 5000).cpu().numpy()
    assert np.array_equal(pos, np.cumsum(gaps)) and gaps.min() >= 1 and gaps.max() <= 49
    part = eng.synth_positions(seed, chrom, 1000, site0=4000).cpu().numpy()
    assert np.array_equal(part, pos[4000:])


def test_fused_site_pass_equals_two_kernels(eng):
    """sai_site_pass (counts kept on chip) == sai_site_counts + sai_site_flags, bit for bit, with
    and without the optional counts output, for 1, 4 and 18 parameter sets (C5's sweep rides in
    the fused pass too) and 1..3 sources."""
    import torch

    from sai_amd import _ffi

    rng = np.random.default_rng(17)
    n_sites = 1500
    n_cand_seen = 0
    for n_src in (1, 3):
        mats = [rng.integers(-2, 3, size=(n_sites, n)).astype(np.int8) for n in [33, 4100, *([2] * n_src)]]
        pl = [2, 2] + [int(rng.integers(1, 4)) for _ in range(n_src)]
        pops = [eng.tile(m) for m in mats]
        for n_sets in (1, 4, 18):
            sets = [
                _ffi.make_params(float(rng.choice([0.1, 0.5, 1.0])), float(rng.choice([0.0, 0.4])), 0.9,
                                 [(str(rng.choice(["=", ">=", "<"])), float(rng.choice([0.0, 0.5, 1.0]))) for _ in range(n_src)],
                                 bool(s % 2))
                for s in range(n_sets)
            ]  # fmt: skip
            counts = eng.site_counts(pops)
            tf, fl, _ = eng.site_flags(counts, pl, sets)
            tf2, fl2 = eng.site_pass(pops, pl, sets)
            c3 = torch.zeros_like(counts)
            tf3, fl3 = eng.site_pass(pops, pl, sets, counts=c3)
            for a, b in ((tf, tf2), (tf, tf3)):
                assert a.cpu().numpy().tobytes() == b.cpu().numpy().tobytes()
            assert torch.equal(fl, fl2) and torch.equal(fl, fl3) and torch.equal(counts, c3)
            # SAI_FREQ_CANDIDATES: same condition / inverted words; "any" = the OR of the conditions instead of
            # all ones, and tgt_freq written for exactly those sites, packed at the start of each tile's slots
            assert bool((fl[:, 0] == -1).all())
            sentinel = -7.25
            out = (torch.full_like(tf, sentinel), torch.zeros_like(fl))
            eng.site_pass(pops, pl, sets, out=out, freq_mode="candidates")
            cand = ((eng.flag_bytes(fl, n_sites, sets) & 1) != 0).any(dim=0)
            assert torch.equal(out[1][:, 1:], fl[:, 1:]) and (int(cand.sum()) < n_sites or n_sets == 18)
            bits = (out[1][:, 0].unsqueeze(1) >> torch.arange(64, device=fl.device)) & 1
            assert torch.equal(bits.reshape(-1)[:n_sites].bool(), cand)
            n_cand_seen += int(cand.sum())
            per_site = eng.site_tgt_freq(out[1], out[0], n_sites)
            assert per_site[cand].cpu().numpy().tobytes() == tf[cand].cpu().numpy().tobytes()
            assert bool(torch.isnan(per_site[~cand]).all())
            n_tiles = (n_sites + 63) // 64
            used = torch.zeros(n_tiles * 64, dtype=torch.bool, device=fl.device)
            used.view(n_tiles, 64)[torch.arange(64, device=fl.device).unsqueeze(0) < bits.sum(dim=1, keepdim=True)] = True
            assert bool((out[0][~used[:n_sites]] == sentinel).all())  # nothing written beyond a tile's packed slots
            tf4, fl4 = eng.site_pass(pops, pl, sets, freq_mode="candidates")  # fresh buffer: NaN elsewhere
            assert torch.equal(fl4, out[1]) and bool(torch.isnan(eng.site_tgt_freq(fl4, tf4, n_sites)[~cand]).all())
    assert n_cand_seen > 0
    with pytest.raises(ValueError, match="at most"):
        eng.site_pass(pops, pl, sets * 2)


@pytest.mark.parametrize("n_sites", [1, 63, 64, 65, 1000])
def test_flag_planes_are_the_oracle_decisions_bit_for_bit(eng, n_sites):
    """The documented row layout (saihip.h): bit b of a word = site 64 t + b; word 0 "any" (all ones from
    a dense pass), word 1 + s compute_matching_loci's condition of set s, word 1 + n + s site inverted (the
    call has a set without ancestral alleles) -- from the stand-alone kernel, the fused pass and the packed2
    pass alike, spare bits of the last tile 0; with ancestral alleles everywhere the inverted words do not
    exist; and flag_bytes' U candidates are the oracle's."""
    from oracle import sai_oracle as O
    from sai_amd import _ffi

    rng = np.random.default_rng(n_sites)
    mats = [rng.integers(-1, 3, size=(n_sites, n)).astype(np.int8) for n in (9, 7, 1, 2)]
    pl = [2, 2, 2, 2]
    specs = [(0.6, 0.3, [("=", 1.0), (">=", 0.5)], True), (0.9, 0.5, [("<=", 0.5), ("=", 0.0)], False),
             (1.0, 0.0, [(">=", 0.0), (">=", 0.0)], False)]  # fmt: skip
    sets = [_ffi.make_params(w, x, 0.9, y, anc) for w, x, y, anc in specs]
    pops = [eng.tile(m) for m in mats]
    tf, planes, _ = eng.site_flags(eng.site_counts(pops), pl, sets)
    assert tuple(planes.shape) == ((n_sites + 63) // 64, 9)
    _, fused = eng.site_pass(pops, pl, sets)
    _, packed = eng.site_pass_packed2([eng.pack2(p) for p in pops], pl, sets)
    words = planes.cpu().numpy().view(np.uint64)
    assert np.array_equal(words, fused.cpu().numpy().view(np.uint64)) and np.array_equal(words, packed.cpu().numpy().view(np.uint64))
    assert np.all(words[:, 0] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(words[:, 7:] == 0)  # 1 + 2 * 3 words used
    site = np.arange(n_sites)
    m64 = [m.astype(np.int64) for m in mats]
    plain = [O.allele_freq(g, 2) for g in m64]
    ok = np.all([np.isfinite(f) & (f >= 0) & (f <= 1) for f in plain], axis=0)
    bytes_ = eng.flag_bytes(planes, n_sites, sets, tf).cpu().numpy()
    for s, (w, x, y, anc) in enumerate(specs):
        _, tfo, cond = O.matching_loci(m64[0], m64[1], m64[2:], w, y, pl, anc)
        mirror = np.all([O._COMPARE[op](f, 1 - yy) for f, (op, yy) in zip(plain[2:], y)], axis=0)
        want = {1 + s: cond, 1 + 3 + s: np.zeros(n_sites, bool) if anc else (mirror & ok)}
        for k, bits in want.items():
            got = (words[site // 64, k] >> (site % 64).astype(np.uint64)) & np.uint64(1)
            assert np.array_equal(got.astype(bool), bits), (s, k)
            if n_sites % 64:  # spare bits of the last tile
                assert int(words[-1, k]) >> (n_sites % 64) == 0
        assert np.array_equal((bytes_[s] & 1).astype(bool), cond) and np.array_equal((bytes_[s] & 4).astype(bool), want[1 + 3 + s])
        assert np.array_equal((bytes_[s] & 2).astype(bool), cond & (tfo > x)), s  # U's candidates (u_statistic.py:92)
    # every set polarised by ancestral alleles: one word per set behind "any", no inverted words
    anc_sets = [_ffi.make_params(w, x, 0.9, y, True) for w, x, y, _ in specs]
    _, p2, _ = eng.site_flags(eng.site_counts(pops), pl, anc_sets)
    _, f2 = eng.site_pass(pops, pl, anc_sets, freq_mode="candidates")
    w2, wf2 = p2.cpu().numpy().view(np.uint64), f2.cpu().numpy().view(np.uint64)
    assert np.all(w2[:, 4:] == 0) and np.array_equal(w2[:, 1:], wf2[:, 1:])
    assert np.array_equal(wf2[:, 0], w2[:, 1] | w2[:, 2] | w2[:, 3])


def test_window_ranges_outside_the_block_are_clamped(eng):
    """lo / hi are the caller's device arrays: a range that reaches outside the block, or runs backwards, is
    clamped to the block (an empty window), never followed into memory that is not there."""
    import torch

    from sai_amd import _ffi

    rng = np.random.default_rng(8)
    n = 1000
    mats = [rng.integers(0, 3, size=(n, k)).astype(np.int8) for k in (6, 5, 1)]
    sets = [_ffi.make_params(1.0, 0.0, 0.5, [(">=", 0.0)], True)]
    pops = [eng.tile(m) for m in mats]
    tgt_freq, planes, _ = eng.site_flags(eng.site_counts(pops), [2, 2, 2], sets)
    lo = torch.tensor([-5, 10, 900, 0, 2_000_000_000, -2_000_000_000], dtype=torch.int32, device=eng.device)
    hi = torch.tensor([3, 5, 5000, 1000, 2_100_000_000, 2_000_000_000], dtype=torch.int32, device=eng.device)
    res = eng.window_stats(tgt_freq, planes, sets, lo, hi)
    assert res.records[0]["n_sites"].tolist() == [3, 0, 100, 1000, 0, 1000]
    cond = (eng.flag_bytes(planes, n, sets)[0].cpu().numpy() & 1).astype(np.int64)
    ranges = [(0, 3), (10, 10), (900, 1000), (0, 1000), (1000, 1000), (0, 1000)]  # what the six clamp to
    assert res.records[0]["n_cond"].tolist() == [int(cond[a:b].sum()) for a, b in ranges] and cond.sum() > 900
    assert all(900 <= i < 1000 for i in res.q_list(0, 2).tolist()) and len(res.q_list(0, 2)) > 0


def test_randomized_windows_against_oracle(eng):
    """Fuzz: random population sizes, ploidies, missing rates, operators, thresholds, window
    grids; every record and candidate list against the per-window oracle."""
    from oracle import sai_oracle as O
    from sai_amd import _ffi

    rng = np.random.default_rng(2026 + int(os.environ.get("SAI_FUZZ_SEED", "0")))  # SAI_FUZZ_SEED: other sequences for the runs at scale
    ops = ["=", "<", ">", "<=", ">="]
    for trial in range(int(os.environ.get("SAI_FUZZ_TRIALS", "25"))):  # 3000 were run once on the GPU box
        n_sites = int(rng.integers(1, 900))
        n_src = int(rng.integers(1, 4))
        sizes = [int(rng.integers(1, 70)), int(rng.integers(1, 70))] + [int(rng.integers(1, 4)) for _ in range(n_src)]
        pl = [int(rng.integers(1, 5)) for _ in range(2 + n_src)]
        miss = float(rng.choice([0.0, 0.01, 0.3]))
        p = rng.random(n_sites) ** float(rng.choice([1, 2, 4]))
        mats = []
        for n, ploidy in zip(sizes, pl):
            g = rng.binomial(ploidy, np.broadcast_to(p[:, None], (n_sites, n))).astype(np.int64)
            if rng.random() < 0.5:
                g[rng.random(n_sites) < 0.3] = ploidy  # fixed sites so "= 1" matches
            g[rng.random(g.shape) < miss] = -ploidy
            mats.append(g)
        pos = np.cumsum(rng.integers(1, 80, n_sites)).astype(np.int64)
        win = int(rng.integers(50, 5000))
        step = int(rng.integers(1, win + 1))
        windows = O.split_windows(pos, win, step)[:400]
        specs = []
        for _ in range(int(rng.integers(1, 4))):
            specs.append(dict(
                w=float(rng.choice([0.0, 0.05, 0.3, 1.0])), x=float(rng.choice([0.0, 0.3, 0.9])),
                quantile=float(rng.choice([0.0, 0.5, 0.95, 1.0, rng.random()])),
                y_list=[(str(rng.choice(ops)), float(rng.choice([0.0, 0.25, 0.5, 1.0]))) for _ in range(n_src)],
                anc=bool(rng.random() < 0.5),
            ))  # fmt: skip
        sets = [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in specs]
        res, lo, hi = _window_pass(eng, mats, pl, sets, pos, np.array([w[0] for w in windows]), np.array([w[1] for w in windows]))
        for si, s in enumerate(specs):
            for wi, (ws, we) in enumerate(windows):
                m = (pos >= ws) & (pos <= we)
                rec = res.records[si, wi]
                assert rec["n_sites"] == int(m.sum())
                if not m.any():
                    assert rec["u_count"] == 0 and np.isnan(rec["q"])
                    continue
                kw = dict(ref_gts=mats[0][m], tgt_gts=mats[1][m], src_gts_list=[g[m] for g in mats[2:]], ref_ploidy=pl[0],
                          tgt_ploidy=pl[1], src_ploidy_list=pl[2:], pos=pos[m], w=s["w"], y_list=s["y_list"],
                          anc_allele_available=s["anc"])  # fmt: skip
                eu = O.u_stat(x=s["x"], **kw)
                eq = O.q_stat(quantile=s["quantile"], **kw)
                assert rec["u_count"] == eu["value"], (trial, si, wi)
                assert res.u_list(si, wi).tolist() == eu["cdd_pos"].tolist()
                assert same_f64(rec["q"], eq["value"]), (trial, si, wi, rec["q"], eq["value"])
                assert res.q_list(si, wi).tolist() == np.asarray(eq["cdd_pos"]).astype(np.int64).tolist()


def test_randomized_stage_shapes_against_oracle(eng):
    """Fuzz of the windows stage's forms and capacities: 1-24 parameter sets (wave form below four, one
    workgroup per window from four on, a second call beyond 20), windows of one site up to thousands (rows
    and stored frequencies inside and beyond what a workgroup keeps in LDS), few individuals (hundreds of
    equal frequencies around the quantile: rank counting, the digit histogram's gather pass and the
    level-by-level select all occur), with and without inverted planes; every record and both lists
    against the per-window oracle."""
    from oracle import sai_oracle as O
    from sai_amd import _ffi

    rng = np.random.default_rng(4242 + int(os.environ.get("SAI_FUZZ_SEED", "0")))
    ops = ["=", "<", ">", "<=", ">="]
    seen_heavy = seen_shared = 0
    for trial in range(int(os.environ.get("SAI_FUZZ_SHAPES", "24"))):  # 6000 were run once on the GPU box
        n_sites = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 3000)), int(rng.integers(3000, 14000))]))
        n_src = int(rng.integers(1, 3))
        sizes = [int(rng.integers(1, 30)), int(rng.choice([1, 3, 17, 60, 200]))] + [int(rng.integers(1, 3)) for _ in range(n_src)]
        pl = [int(rng.integers(1, 5)) for _ in range(2 + n_src)]
        p = rng.random(n_sites) ** float(rng.choice([1, 2, 4]))
        mats = []
        for k, (n, ploidy) in enumerate(zip(sizes, pl)):
            scale = float(rng.choice([0.02, 0.3, 1.0])) if k == 0 else 1.0  # a rare reference makes "ref < w" common
            g = rng.binomial(ploidy, np.broadcast_to((p * scale)[:, None], (n_sites, n))).astype(np.int64)
            if k >= 2 and rng.random() < 0.7:
                g[rng.random(n_sites) < 0.5] = ploidy * int(rng.integers(0, 2))
            g[rng.random(g.shape) < float(rng.choice([0.0, 0.02, 0.3]))] = -ploidy
            mats.append(g)
        pos = np.cumsum(rng.integers(1, int(rng.choice([2, 9, 60])) + 1, n_sites)).astype(np.int64)
        n_win = int(rng.integers(1, 25))
        a = rng.integers(0, n_sites, n_win)
        width = np.minimum(rng.choice([1, 40, 300, 2200, 9000], n_win) * (0.2 + rng.random(n_win)), n_sites).astype(np.int64)
        b = np.minimum(a + width, n_sites - 1)
        ws, we = pos[a] - rng.integers(0, 2, n_win), pos[b] + rng.integers(0, 2, n_win)
        n_sets = int(rng.choice([1, 2, 3, 4, 5, 9, 18, 20, 24]))
        anc_mode = int(rng.integers(0, 3))  # all polarised, none, mixed
        specs = [dict(w=float(rng.choice([0.05, 0.3, 1.0, 1.0])), x=float(rng.choice([0.0, 0.3, 0.9])),
                      quantile=float(rng.choice([0.0, 0.5, 0.95, 1.0, rng.random()])),
                      y_list=[(str(rng.choice(ops)), float(rng.choice([0.0, 0.5, 1.0]))) for _ in range(n_src)],
                      anc=(True, False, bool(rng.random() < 0.5))[anc_mode]) for _ in range(n_sets)]  # fmt: skip
        sets = [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in specs]
        if n_sets <= _ffi.SAI_MAX_SETS:
            res, lo, hi = _window_pass(eng, mats, pl, sets, pos, ws, we)
            get = lambda si: (res, si)  # noqa: E731
        else:  # more sets than one call takes: the callers split them (ResidentScorer does the same)
            m = _ffi.SAI_MAX_SETS
            parts = [_window_pass(eng, mats, pl, sets[i : i + m], pos, ws, we)[0] for i in range(0, n_sets, m)]
            get = lambda si: (parts[si // m], si % m)  # noqa: E731
        seen_shared += n_sets >= 4
        if trial % 2 == 0:
            # the same job through a ResidentScorer (fused pass up to 20 sets, set chunks beyond; plain and
            # pipelined steps): the very records and lists of the two-kernel route above
            import torch

            from sai_amd.resident import ResidentBlock, ResidentScorer

            block = ResidentBlock([eng.tile(g) for g in mats], pl, torch.as_tensor(pos.astype(np.int32)).to(eng.device))
            scorer = ResidentScorer(eng, block, list(zip(ws.tolist(), we.tolist())), sets, cap_u=1 << 10, cap_q=1 << 10,
                                    overlap=trial % 4 == 0)  # fmt: skip
            for _ in range(1 + trial % 3):
                scorer.step()
            got = scorer.results(grow=True)
            for si in range(n_sets):
                r, ri = get(si)
                for f in ("n_sites", "n_cond", "u_count", "n_cdd_q"):
                    assert got.records[si][f].tolist() == r.records[ri][f].tolist(), (trial, si, f)
                assert got.records[si]["q"].tobytes() == r.records[ri]["q"].tobytes(), (trial, si)
                for wi in range(n_win):
                    assert got.u_list(si, wi).tolist() == r.u_list(ri, wi).tolist(), (trial, si, wi)
                    assert got.q_list(si, wi).tolist() == r.q_list(ri, wi).tolist(), (trial, si, wi)
            scorer.close()
        for si, s in enumerate(specs):
            r, ri = get(si)
            for wi in range(n_win):
                msk = (pos >= ws[wi]) & (pos <= we[wi])
                rec = r.records[ri, wi]
                assert rec["n_sites"] == int(msk.sum()), (trial, si, wi)
                if not msk.any():
                    assert rec["u_count"] == 0 and np.isnan(rec["q"])
                    continue
                kw = dict(ref_gts=mats[0][msk], tgt_gts=mats[1][msk], src_gts_list=[g[msk] for g in mats[2:]], ref_ploidy=pl[0],
                          tgt_ploidy=pl[1], src_ploidy_list=pl[2:], pos=pos[msk], w=s["w"], y_list=s["y_list"],
                          anc_allele_available=s["anc"])  # fmt: skip
                eu, eq = O.u_stat(x=s["x"], **kw), O.q_stat(quantile=s["quantile"], **kw)
                seen_heavy += rec["n_cond"] > 256
                assert rec["u_count"] == eu["value"], (trial, si, wi)
                assert r.u_list(ri, wi).tolist() == eu["cdd_pos"].tolist(), (trial, si, wi)
                assert same_f64(rec["q"], eq["value"]), (trial, si, wi, rec["q"], eq["value"])
                assert r.q_list(ri, wi).tolist() == np.asarray(eq["cdd_pos"]).astype(np.int64).tolist(), (trial, si, wi)
    assert seen_heavy > 0 and seen_shared > 0


@pytest.mark.parametrize("n_src", [1, 2, 3, 6])
def test_predicate_table_and_set_by_set_decision_agree(eng, monkeypatch, n_src):
    """The per-site decision from the predicate table (every distinct comparison once, site_eval.hpp) against the
    set-by-set form (SAI_NO_PRED_TABLE=1; also what a call with more than 32 distinct comparisons gets): the same
    planes and stored frequencies from the fused pass and from site_flags, for sweeps like C5's and for random sets
    with all five operators, repeated and unique thresholds, both polarity modes mixed in one row."""
    import torch

    from sai_amd import _ffi

    rng = np.random.default_rng(50 + n_src)
    n_sites = 3000
    ploidy = [2, 3] + [int(rng.integers(1, 4)) for _ in range(n_src)]
    sizes = [40, 30] + [int(rng.integers(1, 4)) for _ in range(n_src)]
    mats = [np.where(rng.random((n_sites, n)) < 0.03, -pl, rng.binomial(pl, rng.random(n_sites)[:, None] ** 2, size=(n_sites, n))).astype(np.int8)
            for n, pl in zip(sizes, ploidy)]  # fmt: skip
    pops = eng.tile_many(mats)
    ops = ["=", "<", ">", "<=", ">="]
    grid = [0.0, 0.25, 1 / 3, 0.5, 2 / 3, 0.75, 1.0]
    for trial in range(6):
        n_sets = int(rng.integers(1, 21))
        few = trial % 2 == 0  # a sweep over a small grid (few distinct comparisons) / every threshold its own
        sets = [_ffi.make_params(float(rng.choice([0.05, 0.3, 0.6]) if few else rng.random()), 0.2, 0.9,
                                 [(str(rng.choice(ops)), float(rng.choice(grid)) if few else float(rng.random())) for _ in range(n_src)],
                                 bool(rng.random() < 0.5)) for _ in range(n_sets)]  # fmt: skip
        got = {}
        for form in ("table", "sets"):
            if form == "sets":
                monkeypatch.setenv("SAI_NO_PRED_TABLE", "1")
            else:
                monkeypatch.delenv("SAI_NO_PRED_TABLE", raising=False)
            counts = eng.site_counts(pops)
            f1, p1, a1 = eng.site_flags(counts, ploidy, sets, want_adj=True)
            f2, p2 = eng.site_pass(pops, ploidy, sets, freq_mode="candidates")
            got[form] = (f1.nan_to_num(-1.0), p1, a1.nan_to_num(-1.0), eng.site_tgt_freq(p2, f2, n_sites).nan_to_num(-1.0), p2)
        for a, b in zip(got["table"], got["sets"]):
            assert torch.equal(a, b), (trial, n_sets)
    monkeypatch.delenv("SAI_NO_PRED_TABLE", raising=False)


@pytest.mark.parametrize("n_sets,polarised", [(20, "none"), (20, "all"), (27, "first chunk only"), (27, "second chunk only"), (45, "mixed")])
def test_rows_of_many_sets_and_both_polarity_modes(eng, n_sets, polarised):
    """A full row (20 sets: 1 + 20 condition words, + 20 inverted words when a set lacks ancestral alleles)
    and sets beyond SAI_MAX_SETS evaluated in several calls whose rows differ in shape -- whether a chunk's
    rows carry inverted words is decided per call, by its own sets, in the call that writes and in the call
    that reads.  Fused pass (<= 20 sets), site_flags + window_stats, and a ResidentScorer; every record and
    list against the oracle."""
    from oracle import sai_oracle as O
    from sai_amd import _ffi
    from sai_amd.resident import ResidentBlock, ResidentScorer

    rng = np.random.default_rng(n_sets * 7 + len(polarised))
    n_sites, sizes, pl = 700, [11, 9, 1, 2], [2, 2, 2, 1]
    p = rng.random(n_sites) ** 2
    mats = []
    for n, ploidy in zip(sizes, pl):
        g = rng.binomial(ploidy, np.broadcast_to(p[:, None], (n_sites, n))).astype(np.int64)
        g[rng.random(n_sites) < 0.25] = ploidy * int(rng.integers(0, 2))
        g[rng.random(g.shape) < 0.02] = -ploidy
        mats.append(g)
    pos = np.cumsum(rng.integers(1, 60, n_sites)).astype(np.int64)
    windows = O.split_windows(pos, 4000, 1500)
    anc_of = {"none": lambda s: False, "all": lambda s: True, "first chunk only": lambda s: s < 20,
              "second chunk only": lambda s: s >= 20, "mixed": lambda s: s % 3 != 0}[polarised]
    ops = ["=", "<", ">", "<=", ">="]
    specs = [dict(w=float(rng.choice([0.3, 0.6, 1.0])), x=float(rng.choice([0.0, 0.2, 0.5])), quantile=float(rng.choice([0.5, 0.9, 1.0])),
                  y_list=[(str(rng.choice(ops)), float(rng.choice([0.0, 0.5, 1.0]))) for _ in range(2)], anc=anc_of(s))
             for s in range(n_sets)]  # fmt: skip
    sets = [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in specs]
    ws, we = np.array([w[0] for w in windows]), np.array([w[1] for w in windows])
    res, lo, hi = _window_pass(eng, mats, pl, sets, pos, ws, we)
    import torch

    pops = [eng.tile(m) for m in mats]
    block = ResidentBlock(pops, pl, torch.as_tensor(pos.astype(np.int32)).to(eng.device))
    scorer = ResidentScorer(eng, block, windows, sets, cap_u=1 << 12, cap_q=1 << 12)
    assert scorer.fused == (n_sets <= _ffi.SAI_FUSED_SETS)
    scorer.step()
    res2 = scorer.results(grow=True)
    for r in (res, res2):
        for si, s in enumerate(specs):
            for wi, (a, b) in enumerate(windows):
                m = (pos >= a) & (pos <= b)
                kw = dict(ref_gts=mats[0][m], tgt_gts=mats[1][m], src_gts_list=[g[m] for g in mats[2:]], ref_ploidy=pl[0],
                          tgt_ploidy=pl[1], src_ploidy_list=pl[2:], pos=pos[m], w=s["w"], y_list=s["y_list"],
                          anc_allele_available=s["anc"])  # fmt: skip
                eu, eq = O.u_stat(x=s["x"], **kw), O.q_stat(quantile=s["quantile"], **kw)
                rec = r.records[si, wi]
                assert rec["u_count"] == eu["value"] and r.u_list(si, wi).tolist() == eu["cdd_pos"].tolist(), (si, wi)
                assert same_f64(rec["q"], eq["value"]), (si, wi)
                assert r.q_list(si, wi).tolist() == np.asarray(eq["cdd_pos"]).astype(np.int64).tolist()
    assert int(res.records["u_count"].sum()) > 0 and np.isfinite(res.records["q"]).any()


@pytest.mark.parametrize("sizes", [[1, 1, 1], [64, 65, 2], [200, 129, 1, 3], [1000, 1008, 2], [4097, 513, 2], [17, 33, 49, 128]])
def test_packed2_layout_equals_int8_path(eng, sizes):
    """The optional 2-bit layout: packing is exact, the packed site pass gives the same counts,
    frequencies and flags as the int8 kernels, and blocks with dosages above 2 are refused."""
    import torch

    from sai_amd import _ffi

    rng = np.random.default_rng(sum(sizes))
    n_sites = 777
    mats = []
    for n in sizes:
        g = rng.integers(0, 3, size=(n_sites, n)).astype(np.int8)
        g[rng.random(g.shape) < 0.03] = -2
        g[rng.random(g.shape) < 0.01] = -1
        mats.append(g)
    mats[0][5] = 2
    mats[1][6] = -1
    pl = [2] * len(sizes)
    tiled = [eng.tile(m) for m in mats]
    packed = [eng.pack2(t) for t in tiled]
    for m, pk in zip(mats, packed):  # the documented bit layout
        n_ind = m.shape[1]
        n_full, w_tail = n_ind // 64, (n_ind % 64 + 15) // 16
        n_tiles = (n_sites + 63) // 64
        tile_words = n_full * 256 + w_tail * 64
        assert pk.data.numel() == n_tiles * tile_words * 4
        words = pk.data.cpu().numpy().view(np.uint32).reshape(n_tiles, tile_words)
        codes = np.where(m < 0, 3, m).astype(np.uint32)
        sites = np.arange(n_sites)

        def field(ind):
            if ind // 64 < n_full:
                w = words[sites // 64, (ind // 64) * 256 + (sites % 64) * 4 + (ind % 64) // 16]
            else:
                w = words[sites // 64, n_full * 256 + (sites % 64) * w_tail + (ind % 64) // 16]
            return (w >> (2 * (ind % 16))) & 3

        for ind in (0, n_ind // 2, n_ind - 1):
            assert np.array_equal(field(ind), codes[:, ind])
        if n_ind % 16:  # padding individuals inside the last word: code 0
            assert not field(n_ind).any()
        if n_sites % 64:  # padding sites of the last tile: code 3 everywhere
            full = words[-1, : n_full * 256].reshape(n_full, 64, 4)[:, n_sites % 64 :, :]
            assert (full == 0xFFFFFFFF).all()
            if w_tail:
                assert (words[-1, n_full * 256 :].reshape(64, w_tail)[n_sites % 64 :] == 0xFFFFFFFF).all()
    n_src = len(sizes) - 2
    sets = [_ffi.make_params(0.4, 0.3, 0.9, [(">=", 0.5)] * n_src, False), _ffi.make_params(1.0, 0.0, 0.5, [("<=", 1.0)] * n_src, True)]
    counts = eng.site_counts(tiled)
    tf, fl, _ = eng.site_flags(counts, pl, sets)
    c2 = torch.zeros_like(counts)
    tf2, fl2 = eng.site_pass_packed2(packed, pl, sets, counts=c2)
    assert torch.equal(counts, c2) and torch.equal(fl, fl2)
    assert tf.cpu().numpy().tobytes() == tf2.cpu().numpy().tobytes()
    c3 = torch.zeros_like(counts)
    eng.site_pass_packed2(packed, pl, [], counts=c3)
    assert torch.equal(counts, c3)
    tf4, fl4 = eng.site_pass_packed2(packed, pl, sets, freq_mode="candidates")
    cand = ((eng.flag_bytes(fl, n_sites, sets) & 1) != 0).any(dim=0)
    per_site = eng.site_tgt_freq(fl4, tf4, n_sites)
    assert torch.equal(fl4[:, 1:], fl[:, 1:]) and per_site[cand].cpu().numpy().tobytes() == tf[cand].cpu().numpy().tobytes()
    assert bool(torch.isnan(per_site[~cand]).all())
    bad = mats[0].copy()
    bad[3, 0] = 3
    with pytest.raises(ValueError, match="dosage above 2"):
        eng.pack2(eng.tile(bad))


def test_stream_read_probe_xor(eng):
    """The bandwidth probe really reads every word: its XOR equals numpy's, for sizes below,
    at and above the 125 KiB run length (whole runs + grid-strided remainder)."""
    import torch

    from sai_amd import _ffi

    rng = np.random.default_rng(5)
    for n_vec in (0, 1, 63, 8000, 8000 * 3 + 77, 8000 * 700 + 5):
        host = rng.integers(0, 2**32, size=n_vec * 4, dtype=np.uint32)
        buf = torch.from_numpy(host.view(np.int32)).to(eng.device)
        out = torch.zeros((1,), dtype=torch.int32, device=eng.device)
        _ffi.check(eng.lib.sai_probe_stream_read(eng.ctx, eng._ptr(buf) if n_vec else None, n_vec * 16, eng._ptr(out), eng._stream()))
        expect = int(np.bitwise_xor.reduce(host)) if n_vec else 0
        assert int(out.cpu().numpy().view(np.uint32)[0]) == expect
    assert eng.probe_stream_read(buf) > 0


@pytest.mark.parametrize("layout,n_sets", [("int8", 2), ("packed2", 1), ("int8", 6)])
def test_overlapped_steps_equal_plain_steps(eng, layout, n_sets):
    """ResidentScorer(overlap=True) pipelines the windows stage of step k under the site pass of
    step k+1 on a second stream with triple-buffered per-site arrays: after 1, 2 and 5 steps its
    records, offsets and candidate lists are byte-identical to the plain scorer's."""
    import torch

    from sai_amd import _ffi
    from sai_amd.resident import ResidentScorer, default_windows, synth_block

    block = synth_block(eng, 77, 3, 300_000, 130, 70, [2], missing_per_million=2000)
    pos = block.pos.cpu().numpy()
    windows = default_windows(int(pos[0]), int(pos[-1]), 50_000, 10_000)
    sets = [_ffi.make_params(0.05 + 0.01 * s, 0.3, 0.9, [("=", 1.0)], bool(s % 2)) for s in range(n_sets)]
    plain = ResidentScorer(eng, block, windows, sets, layout=layout)
    piped = ResidentScorer(eng, block, windows, sets, layout=layout, overlap=True)
    calls = {"plain": [], "piped": []}
    plain.after_stage = calls["plain"].append  # the hook runs once per step, in step order, in both forms
    piped.after_stage = calls["piped"].append
    plain.step()
    want = plain.results()
    assert int(want.records["u_count"].sum()) > 0
    done = 0
    for n_steps in (1, 1, 3):
        for _ in range(n_steps):
            piped.step()
        done += n_steps
        got = piped.results()
        assert got.records.tobytes() == want.records.tobytes(), done
        assert np.array_equal(got.offsets, want.offsets)
        assert np.array_equal(got.cdd_u, want.cdd_u) and np.array_equal(got.cdd_q, want.cdd_q)
        assert torch.equal(piped.flags, plain.flags)
        assert calls["piped"] == list(range(done)) and calls["plain"] == [0]


def test_gather_on_window_stream_with_rccl(eng):
    """bench.py's N>1 step on one GPU: a one-rank RCCL group, the records gathered from inside
    ``scorer.window_stream()`` while the next site pass is already queued on the main stream."""
    import torch
    import torch.distributed as dist

    from sai_amd import _ffi
    from sai_amd.distributed import gather_padded
    from sai_amd.resident import ResidentScorer, default_windows, synth_block

    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    block = synth_block(eng, 78, 2, 200_000, 64, 64, [1])
    pos = block.pos.cpu().numpy()
    windows = default_windows(int(pos[0]), int(pos[-1]), 50_000, 25_000)
    sets = [_ffi.make_params(0.05, 0.3, 0.9, [("=", 1.0)], True)]
    scorer = ResidentScorer(eng, block, windows, sets, overlap=True)
    import socket

    with socket.socket() as sock:  # a free port: the fixed one of an earlier version could be taken
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=eng.device)  # fmt: skip
    try:
        sizes = [scorer.bufs[0].numel()]
        got = None
        got = []
        scorer.after_stage = lambda index: got.append(gather_padded(scorer.bufs[0], sizes))  # on the window stream
        for _ in range(3):
            scorer.step()
        res = scorer.results()
        assert len(got) == 3
        got = got[-1]
        torch.cuda.synchronize()
        assert got is not None and len(got) == 1
        assert got[0].cpu().numpy().tobytes() == res.records.tobytes()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_sites", [1, 2, 63, 64, 65, 66, 4225, 4226, 300_001])
def test_window_bounds_equals_searchsorted(eng, n_sites):
    """The 64-ary wave search: lo = first site with pos >= start, hi = first site with pos > end,
    for windows inside, across and outside the position range, duplicate positions included."""
    import torch

    rng = np.random.default_rng(n_sites)
    pos = np.cumsum(rng.integers(0 if n_sites > 2 else 1, 40, n_sites)).astype(np.int32) + 5  # ties allowed
    top = int(pos[-1])
    starts = np.concatenate([rng.integers(-10, top + 20, 400), [0, 5, int(pos[0]), top, top + 1, int(pos[n_sites // 2])]]).astype(np.int64)
    ends = starts + np.concatenate([rng.integers(-3, 900, 400), [0, 0, 0, 0, 5, 0]]).astype(np.int64)
    lo, hi = eng.window_bounds(torch.as_tensor(pos).to(eng.device), starts, ends)
    want_lo = np.searchsorted(pos, starts, side="left")
    want_hi = np.maximum(np.searchsorted(pos, ends, side="right"), want_lo)
    assert np.array_equal(lo.cpu().numpy(), want_lo) and np.array_equal(hi.cpu().numpy(), want_hi)


def test_window_bounds_forms_agree_also_on_windows_that_end_before_they_start(eng):
    """Both forms of window_bounds -- 32 lanes per bound side by side (up to 32 768 windows), 8 lanes per window
    lo then hi (more; forced here with SAI_BOUNDS_WIDE_MAX=0 in a child process, the knob is read once) --
    against numpy, with many windows whose end lies before their start inside dense positions: hi is never
    below lo."""
    import subprocess
    import sys

    import torch

    from conftest import ROOT

    rng = np.random.default_rng(12)
    n_sites = 50_000
    pos = (np.cumsum(rng.integers(0, 3, n_sites)) + 3).astype(np.int32)  # dense, with ties
    starts = rng.integers(0, int(pos[-1]) + 5, 3000).astype(np.int64)
    ends = starts + rng.integers(-40, 60, 3000)
    want_lo = np.searchsorted(pos, starts, side="left")
    want_hi = np.maximum(np.searchsorted(pos, ends, side="right"), want_lo)
    assert (np.searchsorted(pos, ends, side="right") < want_lo).sum() > 300  # the case is there
    lo, hi = eng.window_bounds(torch.as_tensor(pos).to(eng.device), starts, ends)
    assert np.array_equal(lo.cpu().numpy(), want_lo) and np.array_equal(hi.cpu().numpy(), want_hi)
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from sai_amd.engine import Engine\n"
        "d = np.load(sys.argv[1]); eng = Engine.get(0)\n"
        "lo, hi = eng.window_bounds(torch.as_tensor(d['pos']).to(eng.device), d['starts'], d['ends'])\n"
        "assert np.array_equal(lo.cpu().numpy(), d['lo']) and np.array_equal(hi.cpu().numpy(), d['hi']); print('narrow ok')\n" % str(ROOT)
    )
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        f = td + "/case.npz"
        np.savez(f, pos=pos, starts=starts, ends=ends, lo=want_lo, hi=want_hi)
        res = subprocess.run([sys.executable, "-c", code, f], env={**__import__("os").environ, "SAI_BOUNDS_WIDE_MAX": "0"},
                             capture_output=True, text=True, timeout=600)  # fmt: skip
    assert res.returncode == 0 and "narrow ok" in res.stdout, res.stderr[-2000:]


# ---- np.sum's order in parallel (fourpop.hip: wave_numpy_sum) -------------------------------


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 127, 128, 129, 135, 136, 263, 1000, 2001, 4097, 7688, 7689, 7693, 8191, 8192,
                               8193, 16384 + 7700, 3 * 8192 + 5])  # fmt: skip
def test_pattern_sum_is_np_sum_bit_for_bit(eng, n):
    """calc_pattern_sum on the GPU against numpy itself: the products in population order, the sum
    in np.sum's order (8192-element pieces, pairwise halving, eight running sums per leaf).  Sizes
    7689..8191 are the pieces whose halving tree is seven levels deep (a six-level unrolling summed a
    129..135-element node as one leaf and differed from numpy in the last bits)."""
    from oracle import sai_oracle as O
    from sai_amd.stats import calc_pattern_sum

    rng = np.random.default_rng(n + 1)
    f = [rng.random(n) ** 3 for _ in range(4)]
    for pattern in ("abba", "baba", "bbaa", "baaa", "abaa", "bbbb", "aaab"):
        prod = np.ones_like(f[0])
        for k, c in enumerate(pattern):
            prod *= f[k] if c == "b" else 1 - f[k]
        want = float(np.sum(prod))
        got = calc_pattern_sum(f[0], f[1], f[2], f[3], pattern)
        assert same_f64(got, want), (n, pattern, got.hex(), want.hex())
        assert same_f64(O.numpy_sum(prod), want)  # the oracle's restatement agrees with numpy as well
    with pytest.raises(ValueError, match="four-character"):
        calc_pattern_sum(f[0], f[1], f[2], f[3], "abb")
    with pytest.raises(ValueError, match="Invalid character 'c'"):
        calc_pattern_sum(f[0], f[1], f[2], f[3], "abca")


def test_window_pattern_sums_at_seven_level_sizes(eng):
    """sai_window_fourpop on windows whose site counts need the seven-level tree, several sources,
    NaN sites, against the oracle's four_pop_stats (pinned to the reference's golden capture)."""
    import torch

    from oracle import sai_oracle as O

    rng = np.random.default_rng(77)
    n_sites, n_src = 30_000, 2
    mats = [rng.integers(0, 3, size=(n_sites, k)).astype(np.int8) for k in (9, 7, 2, 3, 5)]  # ref, tgt, s0, s1, out
    mats[1][rng.random(mats[1].shape) < 0.01] = -2
    mats[2][5000] = -2  # an all-missing source site: NaN poisons the windows that hold it
    pops = [eng.tile(m) for m in mats]
    freqs = eng.site_freqs(eng.site_counts(pops), [2] * 5)
    bounds = [(0, 7689), (100, 100 + 8191), (3, 3 + 7700), (0, 30_000), (6000, 6000 + 8192 + 7693), (10, 17), (20, 20)]
    lo = torch.tensor([b[0] for b in bounds], dtype=torch.int32, device=eng.device)
    hi = torch.tensor([b[1] for b in bounds], dtype=torch.int32, device=eng.device)
    got = eng.window_fourpop(freqs, n_src, True, lo, hi).cpu().numpy()
    for wi, (a, b) in enumerate(bounds):
        if b == a:
            continue
        sub = [m[a:b].astype(np.int64) for m in mats]
        want = O.four_pop_stats(sub[0], sub[1], sub[2:4], sub[4], 2, 2, [2, 2], 2)
        for si in range(n_src):
            for k, name in enumerate(("fd", "df", "Danc", "Dplus")):
                assert same_f64(got[wi, si, k], want[name][si]), (wi, si, name, got[wi, si, k], want[name][si])


def test_calc_four_pops_freq_matches_calc_freq(eng):
    from sai_amd.stats import calc_four_pops_freq, calc_freq

    rng = np.random.default_rng(3)
    mats = [rng.integers(-1, 3, size=(300, k)).astype(np.int64) for k in (5, 4, 2, 3)]
    r, t, s, o = calc_four_pops_freq(mats[0], mats[1], mats[2], mats[3], 2, 2, 2, 2)
    for got, m in zip((r, t, s, o), mats):
        want = calc_freq(m, 2)
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])
    r2, t2, s2, o2 = calc_four_pops_freq(mats[0], mats[1], mats[2], ref_ploidy=2, tgt_ploidy=2, src_ploidy=2)
    assert np.array_equal(o2, np.zeros(300)) and np.array_equal(np.nan_to_num(r2), np.nan_to_num(r))


def test_heavy_windows_value_radix_select(eng):
    """The workgroup fallback on many distinct values: large target population with missing calls
    (frequencies k / (2 * called): thousands of distinct rationals, some a few 1e-7 apart, some
    repeated hundreds of times), inversion on and off, every kind of quantile position -- numpy's
    'linear' quantile bit for bit and the same candidate lists."""
    from oracle import sai_oracle as O
    from sai_amd import _ffi

    rng = np.random.default_rng(9)
    n_sites = 9000
    ref = np.zeros((n_sites, 3), dtype=np.int64)
    p = rng.random(n_sites) ** 2
    tgt = rng.binomial(2, p[:, None], size=(n_sites, 700)).astype(np.int64)
    tgt[rng.random(tgt.shape) < 0.1] = -2
    tgt[:2500] = rng.integers(0, 2, size=(2500, 1))  # a block of sites with only a handful of distinct values
    src = np.where(rng.random((n_sites, 1)) < 0.5, 2, 0).astype(np.int64)
    pos = np.arange(1, n_sites + 1, dtype=np.int64) * 2
    ends = [18000, 9000, 5000, 4600, 1500, 12000]
    for anc in (True, False):  # without ancestral alleles the sites with src = 1 are inverted and drop out
        for quantile in (0.0, 0.001, 0.25, 0.5, 0.777, 0.95, 0.999, 1.0):
            sets = [_ffi.make_params(0.5, 0.5, quantile, [(">=", 0.0)], anc)]
            res, _, _ = _window_pass(eng, [ref, tgt, src], [2, 2, 2], sets, pos, np.ones(len(ends), dtype=np.int64), np.array(ends))
            assert res.records[0]["n_cond"].min() > 256, res.records[0]["n_cond"]  # every window takes the workgroup path
            for wi, we in enumerate(ends):
                m = pos <= we
                kw = dict(ref_gts=ref[m], tgt_gts=tgt[m], src_gts_list=[src[m]], ref_ploidy=2, tgt_ploidy=2,
                          src_ploidy_list=[2], pos=pos[m], w=0.5, y_list=[(">=", 0.0)], anc_allele_available=anc)  # fmt: skip
                eq = O.q_stat(quantile=quantile, **kw)
                rec = res.records[0, wi]
                assert same_f64(rec["q"], eq["value"]), (anc, quantile, wi, rec["q"], eq["value"])
                assert rec["n_cdd_q"] == len(eq["cdd_pos"]) and res.q_list(0, wi).tolist() == eq["cdd_pos"].tolist()
