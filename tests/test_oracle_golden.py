"""Pin the numpy oracle to the reference: every golden vector captured by
tests/golden/make_golden.py (reference outputs + the reference tests' own known
answers) must be reproduced exactly -- integers and positions equal, doubles
bit-equal."""

import numpy as np
import pytest

from conftest import load_golden, same_f64, stat_case_inputs, unhex
from oracle import sai_oracle as O

STATS = load_golden("stats_cases.json")


@pytest.mark.parametrize("case", STATS, ids=[c["name"] for c in STATS])
def test_stats_bit_exact(case):
    a = stat_case_inputs(case)
    out = case["out"]
    raw = [O.allele_freq(a["ref_gts"], a["ploidy"][0]), O.allele_freq(a["tgt_gts"], a["ploidy"][1])] + [
        O.allele_freq(s, p) for s, p in zip(a["src_gts_list"], a["ploidy"][2:])
    ]
    for got, exp in zip(raw, out["raw_freq"]):
        assert all(same_f64(g, unhex(e)) for g, e in zip(got, exp))
    rf, tf, cond = O.matching_loci(
        a["ref_gts"], a["tgt_gts"], a["src_gts_list"], a["w"], a["y_list"], a["ploidy"], a["anc"]
    )
    assert all(same_f64(g, unhex(e)) for g, e in zip(rf, out["ref_freq"]))
    assert all(same_f64(g, unhex(e)) for g, e in zip(tf, out["tgt_freq"]))
    assert cond.tolist() == out["condition"]
    kw = dict(
        ref_gts=a["ref_gts"],
        tgt_gts=a["tgt_gts"],
        src_gts_list=a["src_gts_list"],
        ref_ploidy=a["ploidy"][0],
        tgt_ploidy=a["ploidy"][1],
        src_ploidy_list=a["ploidy"][2:],
    )
    u = O.u_stat(**kw, pos=a["pos"], w=a["w"], x=a["x"], y_list=a["y_list"], anc_allele_available=a["anc"])
    assert isinstance(u["value"], int) and u["value"] == out["U"]
    assert u["cdd_pos"].tolist() == out["U_cdd_pos"]
    q = O.q_stat(
        **kw, pos=a["pos"], w=a["w"], y_list=a["y_list"], quantile=a["quantile"], anc_allele_available=a["anc"]
    )
    assert same_f64(q["value"], unhex(out["Q"]))
    assert np.asarray(q["cdd_pos"]).astype(np.int64).tolist() == out["Q_cdd_pos"]
    assert str(q["cdd_pos"].dtype) == out["Q_cdd_dtype"]


def test_linear_quantile_matches_numpy_bitwise():
    rng = np.random.default_rng(7)
    for _ in range(400):
        n = int(rng.integers(1, 40))
        den = int(rng.integers(1, 50))
        vals = np.sort(rng.integers(0, den + 1, size=n) / den)
        for q in (0.0, 0.05, 0.25, 0.5, 0.9, 0.95, 0.99, 1.0, float(rng.random())):
            assert same_f64(O.linear_quantile(vals, q), np.nanquantile(vals, q)), (vals, q)


def test_errors_and_calc_freq():
    g = load_golden("errors_and_freq.json")
    A = np.array
    ref, tgt = A([[0, 1, 0], [1, 1, 0], [0, 0, 1]]), A([[1, 1, 0], [0, 1, 1], [1, 1, 1]])
    srcs = [A([[0, 0, 1], [1, 1, 0], [0, 1, 1]]), A([[1, 1, 0], [1, 0, 0], [1, 1, 0]])]
    y2 = [("=", 0.5), ("=", 0.5)]
    kw = dict(ref_gts=ref, tgt_gts=tgt, src_gts_list=srcs[:1], ref_ploidy=3, tgt_ploidy=1, src_ploidy_list=[2])
    calls = {
        "w_low": lambda: O.matching_loci(ref, tgt, srcs, -0.1, y2, [2, 2, 2], False),
        "w_high": lambda: O.matching_loci(ref, tgt, srcs, 1.1, y2, [2, 2, 2], False),
        "y_low": lambda: O.matching_loci(ref, tgt, srcs, 0.5, [("=", -0.1)], [2, 2, 2], False),
        "y_high": lambda: O.matching_loci(ref, tgt, srcs, 0.5, [("=", 1.1)], [2, 2, 2], False),
        "bad_op": lambda: O.matching_loci(ref, tgt, srcs, 0.5, [("invalid", 0.5)], [2, 2, 2], False),
        "len_mismatch": lambda: O.matching_loci(ref, tgt, srcs, 0.5, [("=", 0.5)], [2, 2, 2], False),
        "ploidy_none": lambda: O.allele_freq(ref, ploidy=None),
        "ploidy_float": lambda: O.allele_freq(ref, ploidy=9.9),
        "ploidy_neg": lambda: O.allele_freq(ref, ploidy=-100),
        "u_missing_kw": lambda: O.u_stat(**kw, pos=A([0, 1, 2]), w=0.5, x=0.5, y_list=[("=", 0)]),
        "q_missing_kw": lambda: O.q_stat(**kw, pos=A([0, 1, 2]), w=0.5, quantile=0.95, anc_allele_available=False),
    }
    for rec in g["errors"]:
        assert rec["exc"] == "ValueError"
        with pytest.raises(ValueError) as ei:
            calls[rec["label"]]()
        assert str(ei.value) == rec["msg"]
    for rec in g["calc_freq"]:
        got = O.allele_freq(A(rec["gts"]), ploidy=rec["ploidy"])
        assert all(same_f64(a, unhex(b)) for a, b in zip(got, rec["freq"]))


def test_window_grid():
    g = load_golden("window_grid.json")
    for c in g["split"]:
        w = O.split_windows(np.array(c["pos"]), c["window_size"], c["step_size"], c["start"])
        assert len(w) == c["n"]
        assert [list(map(int, t)) for t in w[:6]] == c["head"]
        assert [list(map(int, t)) for t in w[-3:]] == c["tail"]
    for c in g["errors"]:
        assert c["msg"] is not None
        with pytest.raises(ValueError) as ei:
            O.split_windows(np.array(c["pos"]), c["window_size"], c["step_size"])
        assert str(ei.value) == c["msg"]
    for c in g["chunks"]:
        w = O.split_windows(np.array(c["pos"]), c["window_size"], c["step_size"])
        assert [list(map(int, t)) for t in O.split_window_ranges(w, c["num_chunks"])] == c["chunks"]


def _validated(stats):
    """What StatConfig's validator leaves behind (stat_config.py:147-157)."""
    out = {}
    for name, prm in stats.items():
        src = {}
        for pop, expr in prm["src"].items():
            op = next(o for o in ("<=", ">=", "=", "<", ">") if o in expr)
            src[pop] = (op, float(expr[len(op):]))
        out[name] = {"ref": prm["ref"], "tgt": prm["tgt"], "src": src}
    return out


PIPE = load_golden("pipeline.json")


@pytest.mark.parametrize("sc", PIPE, ids=[s["name"] for s in PIPE])
def test_pipeline_items_and_text(sc):
    pos = np.array(sc["pos"], dtype=np.int32)
    data = {
        g: {k: O.Chrom(pos, np.array(v, dtype=np.int64).reshape(len(pos), -1)) for k, v in sc["gts"][g].items()}
        for g in ("ref", "tgt", "src")
    }
    stats = _validated(sc["stats"])
    items = O.run_chunk(
        sc["chr_name"],
        data["ref"],
        data["tgt"],
        data["src"],
        sc["win_len"],
        sc["win_step"],
        stats,
        sc["ploidies"],
        sc["anc_allele_available"],
        start=sc["start"],
        end=sc["end"],
    )
    assert len(items) == len(sc["windows"])
    for it, win, exp in zip(items, sc["windows"], sc["items"]):
        assert [it["ref_pop"], it["tgt_pop"], list(it["src_pop_list"]), it["start"], it["end"], it["nsnps"]] == win[:6]
        for k in ("U", "Q"):
            if exp[k] is None:
                assert k not in it
                continue
            if isinstance(exp[k], int):
                assert it[k] == exp[k] and isinstance(it[k], int)
            else:
                assert same_f64(it[k], unhex(exp[k]))
            assert np.asarray(it["cdd_pos"][k]).astype(np.int64).tolist() == exp[f"{k}_cdd"]
    names = list(sc["stats"].keys())
    assert "".join(O.score_lines(items, names)) == sc["text"]["tsv"]
    for k in names:
        assert "".join(O.log_lines(items, k)) == sc["text"][k]


def test_feature_inline_and_example_vcf():
    g = load_golden("feature_inline.json")
    A = np.array
    stats = _validated(
        {
            "U": {"ref": {"ref1": 0.3}, "tgt": {"tgt1": 0.5}, "src": {"src1": "=1", "src2": "=1"}},
            "Q": {"ref": {"ref1": 0.3}, "tgt": {"tgt1": 0.95}, "src": {"src1": "=0.2", "src2": "=0.4"}},
        }
    )
    pl = {"ref": {"ref1": 1}, "tgt": {"tgt1": 1}, "src": {"src1": 1}}
    kw = dict(
        chr_name="21", ref_pop="ref1", tgt_pop="tgt1", src_pop_list=["src1", "src2"], out_pop=None, start=1000,
        end=2000, pos=A([100, 200, 300]), ref_gts=A([[0, 0, 1], [1, 1, 0], [0, 1, 1]]),
        tgt_gts=A([[0, 1, 1], [1, 1, 1], [0, 0, 1]]),
        src_gts_list=[A([[0, 0, 0], [1, 0, 0], [1, 1, 1]]), A([[1, 1, 1], [0, 1, 1], [0, 0, 1]])],
    )  # fmt: skip
    full = O.window_item(stats, pl, False, ploidy_config=pl, **kw)
    assert full["U"] == g["full"]["U"] and same_f64(full["Q"], unhex(g["full"]["Q"]))
    assert full["out_pop"] == "NA" and full["nsnps"] == 3
    none = O.window_item(stats, pl, False, **dict(kw, ref_gts=None, tgt_gts=None, src_gts_list=None))
    assert np.isnan(none["U"]) and np.isnan(none["Q"]) and none["cdd_pos"]["U"].size == 0
    item = {"chr_name": "21", "start": 1000, "end": 2000, "ref_pop": "ref1", "tgt_pop": "tgt1", "out_pop": "NA",
            "src_pop_list": ["src1", "src2"], "nsnps": 10, "U": 5, "Q": 0.8, "cdd_pos": {"U": A([]), "Q": A([])}}  # fmt: skip
    assert O.score_lines([item], ["U", "Q"]) == [g["process_items_line"]]

    ex = load_golden("example_vcf.json")
    gt = np.array(ex["genotypes"]["gt"], dtype=np.int64)
    pos = np.array(ex["genotypes"]["pos"], dtype=np.int32)
    pl = {"ref": {"AFR": 2}, "tgt": {"CHB": 2}, "src": {"Nean": 2}}
    for label in ("q_only", "u_and_q", "example_config"):
        stats = _validated(ex[label]["stats"])
        it = O.window_item(
            stats, pl, False, chr_name="21", ref_pop="AFR", tgt_pop="CHB", src_pop_list=("Nean",), out_pop=None,
            start=1, end=6666, pos=pos, ref_gts=gt[:, 0:5], tgt_gts=gt[:, 5:10], src_gts_list=[gt[:, 10:11]],
            ploidy_config=pl,
        )  # fmt: skip
        names = list(stats.keys())
        assert "".join(O.score_lines([it], names)) == ex[label]["text"]["tsv"]
        for k in names:
            assert "".join(O.log_lines([it], k)) == ex[label]["text"][k]
    assert O.header_line(["U", "Q"]) == "Chrom\tStart\tEnd\tRef\tTgt\tSrc\tOutgroup\tN(Variants)\tU\tQ\n"
    assert O.log_header_line("U") == "Chrom\tStart\tEnd\tU_SNP\n"


# ---- ABBA-BABA family -------------------------------------------------------

import sys  # noqa: E402

sys.path.insert(0, str((__import__("pathlib").Path(__file__).parent / "golden")))
from seeded import fourpop_inputs  # noqa: E402

FOURPOP = load_golden("fourpop_cases.json")


def fourpop_case_inputs(c):
    if "seeded" in c:
        seed, n_sites, n_ref, n_tgt, src_sizes, n_out, miss = c["seeded"]
        return fourpop_inputs(seed, n_sites, n_ref, n_tgt, src_sizes, n_out, c["ploidies"], miss)
    A = lambda v: None if v is None else np.array(v, dtype=np.int64).reshape(len(v), -1)  # noqa: E731
    return A(c["ref_gts"]), A(c["tgt_gts"]), [A(s) for s in c["src_gts_list"]], A(c["out_gts"])


@pytest.mark.parametrize("case", FOURPOP, ids=[c["name"] for c in FOURPOP])
def test_fourpop_bit_exact(case):
    ref, tgt, srcs, out = fourpop_case_inputs(case)
    pl = case["ploidies"]
    got = O.four_pop_stats(ref, tgt, srcs, out, pl[0], pl[1], pl[2], pl[3])
    for k, exp in case["out"].items():
        assert len(got[k]) == len(exp)
        assert all(same_f64(g, unhex(e)) for g, e in zip(got[k], exp)), (k, got[k], exp)


def test_numpy_sum_order_restatement():
    """The summation order the HIP kernel follows == np.sum, bit for bit (incl. > 8192 elements)."""
    rng = np.random.default_rng(5)
    for n in list(range(0, 140)) + [255, 256, 1000, 2001, 8191, 8192, 8193, 12345, 30000]:
        a = rng.random(n) * rng.choice([1e-3, 1.0, 1e3], n)
        assert same_f64(O.numpy_sum(a), np.sum(a)), n
    a = np.array([1.0, np.nan, 2.0] * 50)
    assert np.isnan(O.numpy_sum(a))
    with pytest.raises(ValueError, match="four-character"):
        O.pattern_sum(a, a, a, a, "abb")
    with pytest.raises(ValueError, match="Invalid character"):
        O.pattern_sum(a, a, a, a, "abcx")


PIPE_OUT = load_golden("pipeline_outgroup.json")


def outgroup_scenario_data(sc):
    n_src = sc["n_src"]
    pl = [2, 2, [2] * n_src, 2]
    ref, tgt, srcs, og = fourpop_inputs(sc["seed"], sc["n_sites"], 8, 6, [1] * n_src, 2 if sc["with_out"] else 0, pl, 0.02)
    pos = np.array(sc["pos"], dtype=np.int32)
    return pos, ref, tgt, srcs, og


@pytest.mark.parametrize("sc", PIPE_OUT, ids=[s["name"] for s in PIPE_OUT])
def test_pipeline_with_outgroup_items_and_text(sc):
    pos, ref, tgt, srcs, og = outgroup_scenario_data(sc)
    stats = {}
    for name, prm in sc["stats"].items():
        stats[name] = _validated({name: prm})[name] if name in ("U", "Q") else prm
    items = O.run_chunk(
        "9", {"R": O.Chrom(pos, ref)}, {"T": O.Chrom(pos, tgt)}, {f"S{i}": O.Chrom(pos, s) for i, s in enumerate(srcs)},
        3000, 1500, stats, sc["ploidies"], True, out_data=({"O": O.Chrom(pos, og)} if sc["with_out"] else None),
    )  # fmt: skip
    assert len(items) == sc["n_windows"]
    for it, exp in zip(items, sc["items"]):
        assert it["out_pop"] == exp["out_pop"] and it["nsnps"] == exp["nsnps"]
        for k in ("fd", "df", "Danc", "Dplus", "DD"):
            if k in exp:
                assert all(same_f64(g, unhex(e)) for g, e in zip(it[k], exp[k]))
    names = [n for n, p in sc["stats"].items() if n in ("U", "Q") or p is True]
    assert "".join(O.score_lines(items, names)) == sc["text"]["tsv"]
    for k in ("U", "Q"):
        assert "".join(O.log_lines(items, k)) == sc["text"][k]
    src_pops = list(sc["ploidies"]["src"])
    head = O.header_line(names, src_pops)
    assert head.startswith("Chrom\tStart\tEnd\tRef\tTgt\tSrc\tOutgroup\tN(Variants)\tfd")
    assert ("fd.S0\tfd.S1" in head) == (sc["n_src"] == 2)


# ---- DD ---------------------------------------------------------------------

DD_CASES = load_golden("dd_cases.json")


def dd_case_inputs(c):
    if "seeded" in c:
        seed, n_sites, n_ref, n_tgt, src_sizes, miss = c["seeded"]
        ref, tgt, srcs, _ = fourpop_inputs(seed, n_sites, n_ref, n_tgt, src_sizes, 0, c["ploidies"], miss)
        return ref, tgt, srcs
    A = lambda v: np.array(v, dtype=np.int64)  # noqa: E731
    return A(c["ref_gts"]), A(c["tgt_gts"]), [A(s) for s in c["src_gts_list"]]


@pytest.mark.parametrize("case", DD_CASES, ids=[c["name"] for c in DD_CASES])
def test_dd_bit_exact(case):
    ref, tgt, srcs = dd_case_inputs(case)
    got = O.dd_stat(ref, tgt, srcs)
    assert len(got) == len(case["out"]) and all(same_f64(g, unhex(e)) for g, e in zip(got, case["out"]))


# ---- random chromosomes / configs captured from the reference (make_golden.py section 9) ------

from seeded import fuzz_scenario  # noqa: E402

PIPE_FUZZ = load_golden("pipeline_fuzz.json")


@pytest.mark.parametrize("rec", PIPE_FUZZ, ids=[str(r["seed"]) for r in PIPE_FUZZ])
def test_pipeline_fuzz_text_equals_reference(rec):
    """The oracle's chunk driver + text formatting on 33 random chromosomes / configs (several ref /
    tgt populations, two sources, outgroups, ploidy 1-4, chunk bounds, all seven statistics):
    every byte of the TSV and the two log files as the reference wrote them."""
    sc = fuzz_scenario(rec["seed"])
    pos = sc["pos"]
    ostats = {n: (_validated({n: p})[n] if n in ("U", "Q") else p) for n, p in sc["stats"].items()}
    odata = {grp: {k: O.Chrom(pos, v) for k, v in sc["gts"][grp].items()} for grp in sc["gts"]}
    items = O.run_chunk("7", odata["ref"], odata["tgt"], odata["src"], sc["win"], sc["step"], ostats, sc["pl"], sc["anc"],
                        start=sc["start"], end=sc["end"], out_data=odata["outgroup"] or None)  # fmt: skip
    assert len(items) == rec["n_items"]
    names = list(sc["stats"].keys())
    assert "".join(O.score_lines(items, names)) == rec["text"]["tsv"]
    for k in ("U", "Q"):
        assert "".join(O.log_lines(items, k)) == rec["text"][k]


# ---- more than six source populations (make_golden.py section 11) ------------------------------

from seeded import many_sources_scenario  # noqa: E402

MANY_SOURCES = load_golden("many_sources.json")


@pytest.mark.parametrize("rec", MANY_SOURCES, ids=[f"{r['seed']}-{r['n_src']}src" for r in MANY_SOURCES])
def test_many_sources_text_equals_reference(rec):
    """Seven to ten source populations (the reference loops over any number, stat_utils.py:114-119, 141-152):
    the oracle's chunk driver writes the reference's text, and its U / Q of the whole chromosome as one window
    are the reference's."""
    sc = many_sources_scenario(rec["seed"])
    assert len(sc["gts"]["src"]) == rec["n_src"] >= 7
    pos = sc["pos"]
    ostats = {n: (_validated({n: p})[n] if n in ("U", "Q") else p) for n, p in sc["stats"].items()}
    odata = {grp: {k: O.Chrom(pos, v) for k, v in sc["gts"][grp].items()} for grp in sc["gts"]}
    items = O.run_chunk("7", odata["ref"], odata["tgt"], odata["src"], sc["win"], sc["step"], ostats, sc["pl"], sc["anc"],
                        out_data=odata["outgroup"] or None)  # fmt: skip
    assert len(items) == rec["n_items"]
    names = list(sc["stats"].keys())
    assert "".join(O.score_lines(items, names)) == rec["text"]["tsv"]
    for k in ("U", "Q"):
        assert "".join(O.log_lines(items, k)) == rec["text"][k]
    t0 = rec["whole"]["tgt"]
    kw = dict(ref_gts=sc["gts"]["ref"]["R0"], tgt_gts=sc["gts"]["tgt"][t0], src_gts_list=list(sc["gts"]["src"].values()),
              ref_ploidy=2, tgt_ploidy=sc["pl"]["tgt"][t0], src_ploidy_list=list(sc["pl"]["src"].values()), pos=pos,
              anc_allele_available=sc["anc"])  # fmt: skip
    u = O.u_stat(w=ostats["U"]["ref"]["R0"], x=ostats["U"]["tgt"][t0], y_list=list(ostats["U"]["src"].values()), **kw)
    q = O.q_stat(w=ostats["Q"]["ref"]["R0"], quantile=ostats["Q"]["tgt"][t0], y_list=list(ostats["Q"]["src"].values()), **kw)
    assert u["value"] == rec["whole"]["U"] and np.asarray(u["cdd_pos"]).astype(np.int64).tolist() == rec["whole"]["U_cdd_pos"]
    assert same_f64(q["value"], unhex(rec["whole"]["Q"])) and np.asarray(q["cdd_pos"]).astype(np.int64).tolist() == rec["whole"]["Q_cdd_pos"]


# ---- populations with different site sets / repeated positions (make_golden.py section 10) ----

SITESETS = load_golden("sitesets.json")


def siteset_inputs(case):
    from seeded import siteset_scenario

    return siteset_scenario(case["kind"], case["seed"])


@pytest.mark.parametrize("case", SITESETS, ids=[f"{c['kind']}-{c['seed']}" for c in SITESETS])
def test_oracle_on_ragged_and_repeated_positions(case):
    """intersect1d + isin per window (window_generator.py:193-231): the oracle writes the reference's
    text when populations lack sites, and fails with the reference's exception type when a position
    is repeated (IndexError from `pos[idx]` / `pos[condition]`, ValueError when only some
    populations repeat it)."""
    sc = siteset_inputs(case)
    data = {g: {k: O.Chrom(sc["pos"][g][k], v) for k, v in sc["gts"][g].items()} for g in ("ref", "tgt", "src")}
    stats = _validated(sc["stats"])
    names = list(sc["stats"])
    if "error" in case:
        with pytest.raises({"IndexError": IndexError, "ValueError": ValueError}[case["error"][0]]):
            O.run_chunk("5", data["ref"], data["tgt"], data["src"], sc["win"], sc["step"], stats, sc["pl"], sc["anc"])
        return
    items = O.run_chunk("5", data["ref"], data["tgt"], data["src"], sc["win"], sc["step"], stats, sc["pl"], sc["anc"])
    assert len(items) == case["n_items"]
    assert "".join(O.score_lines(items, names)) == case["text"]["tsv"]
    for k in names:
        assert "".join(O.log_lines(items, k)) == case["text"][k]
