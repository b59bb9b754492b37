"""CPU tests of the host side of sai_amd: window grid, chunking, VCF ingest and polarisation,
configs, generators, output formatting -- against the golden vectors and the pins the
reference's own tests hold (cited per test).  No GPU is touched."""

import json
import os

import numpy as np
import pytest

from conftest import DATA, load_golden, unhex


# ---- window grid (a9) --------------------------------------------------------------------


def test_split_genome_golden():
    from sai_amd.utils import split_genome, split_windows_ranges

    g = load_golden("window_grid.json")
    for c in g["split"]:
        w = split_genome(np.array(c["pos"]), c["window_size"], c["step_size"], c["start"])
        assert len(w) == c["n"]
        assert [list(t) for t in w[:6]] == c["head"] and [list(t) for t in w[-3:]] == c["tail"]
    for c in g["errors"]:
        with pytest.raises(ValueError) as ei:
            split_genome(np.array(c["pos"]), c["window_size"], c["step_size"])
        assert str(ei.value) == c["msg"]
    for c in g["chunks"]:
        w = split_genome(np.array(c["pos"]), c["window_size"], c["step_size"])
        assert [list(t) for t in split_windows_ranges(w, c["num_chunks"])] == c["chunks"]
    # reference tests/utils/test_utils.py:423-431
    assert split_genome(np.arange(0, 101, 10), 30, 20) == [(1, 30), (21, 50), (41, 70), (61, 90), (81, 110)]


def test_split_genome_matches_oracle_on_a_sweep():
    from oracle import sai_oracle as O
    from sai_amd.utils import split_genome

    rng = np.random.default_rng(0)
    for _ in range(300):
        a = int(rng.integers(0, 10**6))
        b = a + int(rng.integers(0, 10**5))
        step = int(rng.integers(1, 5000))
        win = step + int(rng.integers(0, 20000))
        start = None if rng.random() < 0.5 else int(rng.integers(1, a + 2))
        assert split_genome([a, b], win, step, start) == O.split_windows([a, b], win, step, start)


def test_chunk_generator_pins(in_repo_root):
    # reference tests/generators/test_chunk_generator.py:25-51
    from sai_amd.generators import ChunkGenerator

    g = ChunkGenerator(vcf_file="tests/data/test.data.vcf", chr_name="21", step_size=5000, window_size=10000, num_chunks=2)
    assert len(g) == 2 and g.chunks == [(1, 30000), (25001, 55000)]
    assert list(g.get()) == [{"chr_name": "21", "start": 1, "end": 30000}, {"chr_name": "21", "start": 25001, "end": 55000}]
    with pytest.raises(ValueError, match="Chromosome 1 not found in VCF."):
        ChunkGenerator(vcf_file="tests/data/test.data.vcf", chr_name="1", step_size=10000, window_size=10000, num_chunks=2)


# ---- ingest (a14, minimal) ---------------------------------------------------------------


def test_vcf_reader_example_matches_golden_genotypes(in_repo_root):
    from sai_amd.utils import parse_ind_file
    from sai_amd.utils.vcf import dosage_matrices, read_region

    ex = load_golden("example_vcf.json")["genotypes"]
    names = [f"ind{i}" for i in range(1, 12)]
    reg = read_region("tests/data/example.vcf", "21", names)
    assert reg.pos.tolist() == ex["pos"] and reg.pos.dtype == np.int32
    dos, fdos = dosage_matrices(reg, list(range(11)), 2)
    assert dos.dtype == np.int8 and dos.tolist() == ex["gt"]
    assert dos[11, 0] == -2 and fdos[11, 0] == 4  # ".|." flips to 2+2, as abs(g - 1) does (utils.py:555)
    assert parse_ind_file("tests/data/example.ref.ind.list") == {"AFR": ["ind1", "ind2", "ind3", "ind4", "ind5"]}
    sub = read_region("tests/data/example.vcf", "21", ["ind11", "ind6"], start=222, end=999)
    assert sub.pos.tolist() == [222, 333, 444, 555, 666, 777, 888, 999] and sub.gt[0] == ["1|1", "0|1"]
    with pytest.raises(ValueError, match="samples not found"):
        read_region("tests/data/example.vcf", "21", ["nobody"])


def test_read_data_and_polarisation_pins(in_repo_root):
    """tests/utils/test_utils.py:204-209 (BED), :269-318 (polarised genotypes and positions)."""
    from sai_amd.configs import PloidyConfig
    from sai_amd.utils import read_anc_allele
    from sai_amd.utils import read_dosage_data as read_data

    assert read_anc_allele("tests/data/test.anc.allele.bed", "21") == {"21": {2309: "G", 7879: "A", 11484: "-", 48989: "C"}}
    pc = PloidyConfig({"ref": {"ref1": 2}, "tgt": {"tgt1": 2, "tgt2": 2}, "src": {"src1": 2, "src2": 2}})
    plain = read_data("tests/data/test.data.vcf", "21", pc, "tests/data/test.ref.ind.list", "tests/data/test.tgt.ind.list", None)
    assert plain["src"] == (None, None) and plain["outgroup"] == (None, None)
    assert len(plain["ref"][0]["ref1"].POS) == 19 and plain["ref"][0]["ref1"].GT.shape == (19, 2)
    assert plain["tgt"][1] == {"tgt1": ["ind1", "ind2"], "tgt2": ["ind3", "ind4"]}
    pol = read_data("tests/data/test.data.vcf", "21", pc, "tests/data/test.ref.ind.list", "tests/data/test.tgt.ind.list",
                    None, anc_allele_file="tests/data/test.anc.allele.bed")  # fmt: skip
    exp_pos = [2309, 7879, 48989]
    # expected phased calls of the reference test, summed over the ploidy axis
    assert pol["tgt"][0]["tgt1"].POS.tolist() == exp_pos and pol["tgt"][0]["tgt2"].POS.tolist() == exp_pos
    assert pol["ref"][0]["ref1"].GT.tolist() == [[0, 0], [2, 2], [0, 0]]
    assert pol["tgt"][0]["tgt1"].GT.tolist() == [[1, 0], [2, 1], [0, 1]]
    assert pol["tgt"][0]["tgt2"].GT.tolist() == [[0, 0], [2, 2], [0, 0]]
    with pytest.raises(ValueError, match="No ancestral allele is found for chromosome 21 in the region"):
        read_anc_allele("tests/data/test.anc.allele.bed", "21", start=100, end=200)
    with pytest.raises(ValueError, match="not found in sample file"):
        read_data("tests/data/test.data.vcf", "21", PloidyConfig({"ref": {"nope": 2}, "tgt": {"tgt1": 2}, "src": {"s": 2}}),
                  "tests/data/test.ref.ind.list", "tests/data/test.tgt.ind.list", None)  # fmt: skip


def test_mixed_ploidy_gz_ingest(in_repo_root):
    from sai_amd.configs import PloidyConfig
    from sai_amd.utils import read_dosage_data as read_data

    pc = PloidyConfig({"ref": {"ref1": 2}, "tgt": {"tgt1": 4, "tgt2": 4}, "src": {"src1": 4, "src2": 4}})
    d = read_data("tests/data/test.mixed.ploidy.data.vcf.gz", "21", pc, "tests/data/test.ref.ind.list",
                  "tests/data/test.tgt.ind.list", "tests/data/test.src.ind.list")  # fmt: skip
    assert d["tgt"][0]["tgt1"].GT[0].tolist() == [2, 0]  # 1|0|1|0 and 0|0|0|0
    assert d["ref"][0]["ref1"].GT[3].tolist() == [0, 1] and d["src"][0]["src2"].GT[3].tolist() == [4]


# ---- generators (a8) ---------------------------------------------------------------------


def test_window_generator_counts_and_none_branch(in_repo_root):
    # reference tests/generators/test_window_generator.py:58-97 (380 windows; None generator)
    from sai_amd.configs import PloidyConfig
    from sai_amd.generators import WindowGenerator

    pc = PloidyConfig({"ref": {"ref1": 2}, "tgt": {"tgt1": 2, "tgt2": 2}, "src": {"src1": 2, "src2": 2}})
    kw = dict(vcf_file="tests/data/test.data.vcf", chr_name="21", ref_ind_file="tests/data/test.ref.ind.list",
              tgt_ind_file="tests/data/test.tgt.ind.list", src_ind_file="tests/data/test.src.ind.list", out_ind_file=None,
              win_len=1000, win_step=500, ploidy_config=pc)  # fmt: skip
    g = WindowGenerator(**kw)
    items = list(g.get())
    assert len(items) == 380 == len(g)
    first = items[0]
    assert first["ref_pop"] == "ref1" and set(first) == {
        "chr_name", "ref_pop", "tgt_pop", "src_pop_list", "out_pop", "start", "end", "pos", "ref_gts", "tgt_gts",
        "src_gts_list", "out_gts", "ploidy_config"}  # fmt: skip
    g.ref_data = None
    none_items = list(g.get())
    assert len(none_items) == 380 and none_items[0]["ref_gts"] is None and len(none_items[0]["pos"]) == 0
    assert none_items[0]["ploidy_config"].get_ploidy("src")[0] == 2
    g2 = WindowGenerator(**dict(kw, num_src=2))
    items2 = list(g2.get())
    assert len(items2) == 190 and all(len(i["src_pop_list"]) == 2 for i in items2)
    with pytest.raises(ValueError, match="`win_len` must be greater than 0."):
        WindowGenerator(**dict(kw, win_len=0))


PIPE = load_golden("pipeline.json")


def _from_scenario(sc):
    from sai_amd.configs import PloidyConfig
    from sai_amd.generators import WindowGenerator
    from sai_amd.utils import ChromosomeData

    pos = np.array(sc["pos"], dtype=np.int32)
    start, end = sc["start"], sc["end"]
    sel = np.ones(len(pos), bool) if start is None else (pos >= start) & (pos <= end)
    data = {
        g: {k: ChromosomeData(pos[sel], None, None, np.array(v, dtype=np.int8).reshape(len(pos), -1)[sel])
            for k, v in sc["gts"][g].items()}
        for g in ("ref", "tgt", "src")
    }  # fmt: skip
    return WindowGenerator.from_arrays(
        sc["chr_name"], data["ref"], data["tgt"], data["src"], sc["win_len"], sc["win_step"],
        PloidyConfig(sc["ploidies"]), start=start, end=end,
    ), pos  # fmt: skip


@pytest.mark.parametrize("sc", PIPE, ids=[s["name"] for s in PIPE])
def test_window_generator_equals_reference_sequence(sc):
    """Same ordered (ref, tgt, src, start, end, nsnps, first site) sequence as the reference's
    WindowGenerator produced on the same resident arrays."""
    wg, pos = _from_scenario(sc)
    got = []
    for it in wg.get():
        p = it["pos"]
        lo = int(np.searchsorted(pos, p[0])) if len(p) else -1
        got.append([it["ref_pop"], it["tgt_pop"], list(it["src_pop_list"]), int(it["start"]), int(it["end"]), len(p), lo])
        if len(p):
            assert it["ref_gts"].shape[0] == len(p) == it["tgt_gts"].shape[0]
    assert got == sc["windows"]


# ---- configs (a12) -----------------------------------------------------------------------


def test_configs_behaviour(in_repo_root):
    from pydantic import ValidationError

    from sai_amd.configs import GlobalConfig, PloidyConfig, PopConfig, StatConfig

    sc = StatConfig({"U": {"ref": {"a": 0.01}, "tgt": {"b": 0.5}, "src": {"n": "=1", "d": ">=0.8"}}, "fd": False})
    assert sc.get_parameters("U")["src"] == {"n": ("=", 1.0), "d": (">=", 0.8)}
    assert sc.get_parameters("Q") is None and sc.get_parameters("fd") is False
    for bad in (
        {"X": True},
        {"U": {"ref": {"a": 0.1}, "tgt": {"b": 0.5}}},
        {"U": {"ref": {"a": 1.5}, "tgt": {"b": 0.5}, "src": {"n": "=1"}}},
        {"U": {"ref": {"a": 0.5}, "tgt": {"b": 0.5}, "src": {"n": "1"}}},
        {"Q": {"ref": {"a": 0.5}, "tgt": {"b": 0.5}, "src": {"n": "=1.5"}}},
        {"Q": {"ref": {"a": 0.5}, "tgt": {"b": 0.5}, "src": {"n": "=abc"}}},
    ):
        with pytest.raises(ValueError):
            StatConfig(bad)
    pc = PloidyConfig({"ref": {"a": 2}, "tgt": {"b": 4}, "src": {"n": 2, "d": 1}})
    assert pc.get_ploidy("src") == [2, 1] and pc.get_ploidy("tgt", "b") == 4 and pc.get_ploidy("outgroup") is None
    with pytest.raises(KeyError):
        pc.get_ploidy("ref", "zzz")
    with pytest.raises(ValidationError, match="Missing required ploidy keys"):
        PloidyConfig({"ref": {"a": 2}, "tgt": {"b": 2}})
    with pytest.raises(ValidationError, match="Unsupported ploidy keys"):
        PloidyConfig({"ref": {"a": 2}, "tgt": {"b": 2}, "src": {"c": 2}, "zzz": {"c": 2}})
    with pytest.raises(ValidationError, match="must be a positive integer"):
        PloidyConfig({"ref": {"a": 0}, "tgt": {"b": 2}, "src": {"c": 2}})
    with pytest.raises(ValueError, match="Missing required population keys"):
        PopConfig({"ref": "tests/data/example.ref.ind.list"})
    with pytest.raises(ValueError, match="does not exist"):
        PopConfig({"ref": "nope", "tgt": "nope", "src": "nope"})
    import yaml

    from sai_amd.utils import UniqueKeyLoader

    cfg = yaml.load(open("tests/data/example.u_and_q.config.yaml"), Loader=UniqueKeyLoader)
    gc = GlobalConfig(**cfg)
    assert list(gc.statistics.root) == ["U", "Q"] and gc.populations.get_population("outgroup") is None
    with pytest.raises(ValueError, match="Missing required fields in configuration: ploidies"):
        GlobalConfig(**{k: v for k, v in cfg.items() if k != "ploidies"})
    bad = json.loads(json.dumps(cfg))
    bad["statistics"]["Q"]["tgt"] = {"popB": 0.9}
    with pytest.raises(ValueError, match=r"Population 'popB' used in statistics\[Q\]\[tgt\] is not defined in ploidies\[tgt\]"):
        GlobalConfig(**bad)
    with pytest.raises(ValueError, match="Duplicate key in YAML"):
        yaml.load("a: 1\na: 2\n", Loader=UniqueKeyLoader)


# ---- output (a10) ------------------------------------------------------------------------


def _items_from_golden(sc):
    names = list(sc["stats"].keys())
    items = []
    for win, g in zip(sc["windows"], sc["items"]):
        it = {"chr_name": sc["chr_name"], "ref_pop": win[0], "tgt_pop": win[1], "src_pop_list": tuple(win[2]),
              "start": win[3], "end": win[4], "out_pop": "NA", "nsnps": g["nsnps"], "cdd_pos": {}}  # fmt: skip
        for k in names:
            v = g[k]
            it[k] = v if isinstance(v, int) else np.float64(unhex(v))
            if isinstance(v, str) and v == "nan":
                it[k] = np.nan
            it["cdd_pos"][k] = np.array(g[f"{k}_cdd"], dtype=np.int32) if g[f"{k}_cdd"] else np.array([])
        items.append(it)
    return items, names


@pytest.mark.parametrize("sc", PIPE, ids=[s["name"] for s in PIPE])
def test_process_items_text_equals_reference(sc, tmp_path):
    from sai_amd.configs import PloidyConfig, StatConfig
    from sai_amd.preprocessors import FeaturePreprocessor
    from sai_amd.sai import write_headers

    items, names = _items_from_golden(sc)
    out = tmp_path / "sub" / "o.tsv"
    stat_config = StatConfig(json.loads(json.dumps(sc["stats"])))
    write_headers(str(out), stat_config, PloidyConfig(sc["ploidies"]))
    fp = FeaturePreprocessor(str(out), stat_config, sc["anc_allele_available"])
    fp.process_items(items)
    head = "Chrom\tStart\tEnd\tRef\tTgt\tSrc\tOutgroup\tN(Variants)\t" + "\t".join(names) + "\n"
    assert out.read_text() == head + sc["text"]["tsv"]
    for k in names:
        assert (tmp_path / "sub" / f"o.{k}.log").read_text() == f"Chrom\tStart\tEnd\t{k}_SNP\n" + sc["text"][k]


def test_process_items_reference_line(tmp_path):
    # reference tests/preprocessors/test_feature_preprocessor.py:128-159
    from sai_amd.configs import StatConfig
    from sai_amd.preprocessors import FeaturePreprocessor

    sc = StatConfig({"DD": False,
                     "U": {"ref": {"ref1": 0.3}, "tgt": {"tgt1": 0.5}, "src": {"src1": "=1", "src2": "=1"}},
                     "Q": {"ref": {"ref1": 0.3}, "tgt": {"tgt1": 0.95}, "src": {"src1": "=0.2", "src2": "=0.4"}}})  # fmt: skip
    out = tmp_path / "t.tsv"
    fp = FeaturePreprocessor(str(out), sc)
    item = {"chr_name": "21", "start": 1000, "end": 2000, "ref_pop": "ref1", "tgt_pop": "tgt1", "out_pop": "NA",
            "src_pop_list": ["src1", "src2"], "nsnps": 10, "U": 5, "Q": 0.8, "cdd_pos": {"U": np.array([]), "Q": np.array([])}}  # fmt: skip
    fp.process_items([item])
    assert out.read_text() == load_golden("feature_inline.json")["process_items_line"]
    assert (tmp_path / "t.U.log").read_text() == "21\t1000\t2000\tNA\n"
    assert FeaturePreprocessor(str(out), StatConfig({"DD": True, "fd": False}))._active_stats() == ["DD"]


def test_score_config_errors_without_gpu(in_repo_root, tmp_path):
    # reference tests/test_sai.py:66-89, 113-124
    from sai_amd.sai import score

    kw = dict(vcf_file="tests/data/example.vcf", chr_name="21", win_len=6666, win_step=6666, anc_allele_file=None,
              output_file=str(tmp_path / "o.tsv"), num_workers=1)  # fmt: skip
    with pytest.raises(FileNotFoundError, match="not found"):
        score(config="config.yaml", **kw)
    with pytest.raises(ValueError, match="Error parsing YAML configuration file"):
        score(config="tests/data/invalid.yaml", **kw)
    with pytest.raises(ValueError, match="requires polarized data"):
        score(config="tests/data/test_mixed_ploidy.config.yaml", **kw)


def test_score_takes_the_preloaded_blocks_only_when_the_chunk_holds_them(in_repo_root, tmp_path, monkeypatch):
    """Host logic of ``score``'s one-pass order for a plain-text file (sai._scan_while_reading), with the
    GPU reader stood in for: the scan's span lays out the chunk; the blocks read meanwhile are handed to
    ``run_and_write`` (the one chunk of a one-process run: scored and written in one call) when every kept
    position lies inside the chunk, not otherwise; a read that failed is repeated in the usual order; a
    chromosome the scan does not find is ChunkGenerator's error; a chromosome cut into several chunks (the HBM
    budget) goes chunk by chunk through ``run_compact`` and is written at the end."""
    from sai_amd import sai as sai_mod
    from sai_amd.preprocessors import ChunkPreprocessor

    calls = []
    monkeypatch.setattr(sai_mod, "_reads_in_one_pass", lambda vcf: True)
    monkeypatch.setattr(ChunkPreprocessor, "write_results", lambda self, batches: calls.append(("write", len(batches))))
    monkeypatch.setattr(ChunkPreprocessor, "run_compact",
                        lambda self, chr_name, start, end, preloaded=None: calls.append(("compact", start, end, preloaded)) or "batch")  # fmt: skip
    monkeypatch.setattr(ChunkPreprocessor, "run_and_write",
                        lambda self, chr_name, start, end, preloaded=None: calls.append(("run", start, end, preloaded)))  # fmt: skip
    monkeypatch.delenv("SAI_AMD_HBM_BUDGET_BYTES", raising=False)
    kw = dict(vcf_file="tests/data/test.data.vcf", win_len=10000, win_step=5000, anc_allele_file=None,
              output_file=str(tmp_path / "o.tsv"), config="tests/data/test.uq.config.yaml", num_workers=1)  # fmt: skip

    def preload_with(span):
        def preload(self, chr_name):
            if span == "fails":
                raise ValueError("reader says no")
            return {"ref": "blocks"}, "pos_dev", span
        return preload

    for span, want in (((2309, 48989), ({"ref": "blocks"}, "pos_dev")), (None, ({"ref": "blocks"}, "pos_dev")),
                       ((2309, 200_000), None), ((1, 48989), ({"ref": "blocks"}, "pos_dev")), ((0, 48989), None), ("fails", None)):  # fmt: skip
        calls.clear()
        monkeypatch.setattr(ChunkPreprocessor, "preload", preload_with(span))
        sai_mod.score(chr_name="21", **kw)
        assert calls == [("run", 1, 55000, want)], span  # chr 21: 2309 .. 48989 -> windows 1 .. 55000
    with pytest.raises(ValueError, match="Chromosome 9 not found in VCF"):
        sai_mod.score(chr_name="9", **kw)
    monkeypatch.setattr(sai_mod, "_reads_in_one_pass", lambda vcf: False)  # compressed file / no GPU: scan, then read
    calls.clear()
    sai_mod.score(chr_name="21", **kw)
    assert calls == [("run", 1, 55000, None)]
    monkeypatch.setenv("SAI_AMD_HBM_BUDGET_BYTES", "700")  # 5 509 bytes of text = 1 377 resident bytes: two chunks
    calls.clear()
    sai_mod.score(chr_name="21", **kw)
    assert [c[0] for c in calls] == ["compact", "compact", "write"] and calls[-1] == ("write", 2)
    assert calls[0][1] == 1 and calls[1][2] == 55000 and calls[0][2] + 1 - 10000 + 5000 == calls[1][1]  # window-aligned cut


def test_cli_parser(in_repo_root):
    import argparse

    from sai_amd.__main__ import _sai_cli_parser
    from sai_amd.parsers.argument_validation import existed_file, positive_int

    args = _sai_cli_parser().parse_args(
        ["score", "--vcf", "tests/data/example.vcf", "--chr-name", "21", "--output", "o.tsv", "--config",
         "tests/data/test_sai.config.yaml"]  # fmt: skip
    )
    assert (args.win_len, args.win_step, args.anc_alleles, args.chr_name) == (50000, 10000, None, "21")
    with pytest.raises(argparse.ArgumentTypeError, match="0 is not a positive integer"):
        positive_int("0")
    with pytest.raises(argparse.ArgumentTypeError, match="abc is not a valid integer"):
        positive_int("abc")
    with pytest.raises(argparse.ArgumentTypeError, match="non_existent_file is not found"):
        existed_file("non_existent_file")


def test_registry_behaviour():
    # reference tests/registries/test_registries.py
    import sai_amd.stats  # noqa: F401
    from sai_amd.registries import STAT_REGISTRY, GenericRegistry
    from sai_amd.stats import QStatistic, UStatistic

    assert STAT_REGISTRY.get("U") is UStatistic and STAT_REGISTRY.get("Q") is QStatistic
    assert sorted(STAT_REGISTRY.list_registered()) == ["DD", "Danc", "Dplus", "Q", "U", "df", "fd"]
    with pytest.raises(KeyError, match="No component registered under name 'nope'"):
        STAT_REGISTRY.get("nope")

    class R(GenericRegistry):
        pass

    r = R()
    r.register("a")(int)
    with pytest.raises(ValueError, match="'a' is already registered."):
        r.register("a")(float)


def test_rows_per_statistic_lays_merged_sets_out_per_name():
    """score_windows answers U and Q from ONE parameter set when they share w / sources / polarity; the
    batch still carries one row per configured statistic, each row's lists behind each other."""
    from sai_amd.engine import RECORD_DTYPE, WindowResults
    from sai_amd.preprocessors.feature_preprocessor import _rows_per_statistic

    rng = np.random.default_rng(3)
    n_w = 7
    rec = np.zeros((2, n_w), dtype=RECORD_DTYPE)
    rec["u_count"] = rng.integers(0, 4, (2, n_w))
    rec["n_cdd_q"] = rng.integers(0, 3, (2, n_w))
    rec["q"] = rng.random((2, n_w))
    off = np.zeros((2, n_w, 2), dtype=np.int64)
    for k, name in enumerate(("u_count", "n_cdd_q")):
        flat = rec[name].reshape(-1).astype(np.int64)
        off[:, :, k] = (np.cumsum(flat) - flat).reshape(2, n_w)
    res = WindowResults(rec, off, np.arange(int(rec["u_count"].sum()), dtype=np.int32) + 100,
                        np.arange(int(rec["n_cdd_q"].sum()), dtype=np.int32) + 900)  # fmt: skip
    assert _rows_per_statistic(res, [0, 1]) is res
    for set_of in ([0, 0], [1, 0], [1, 1, 0]):
        shared = _rows_per_statistic(res, set_of)
        assert shared.shared_lists and shared.cdd_u is res.cdd_u and shared.records.shape == (len(set_of), n_w)
        for i, s_i in enumerate(set_of):  # local consumers read through the offsets: the same lists
            for w in range(n_w):
                assert shared.u_list(i, w).tolist() == res.u_list(s_i, w).tolist()
                assert shared.q_list(i, w).tolist() == res.q_list(s_i, w).tolist()
        out = shared.separate_lists()  # the form that travels
        assert not out.shared_lists and out.records.shape == (len(set_of), n_w)
        assert out.cdd_u.size == int(out.records["u_count"].sum()) and out.cdd_q.size == int(out.records["n_cdd_q"].sum())
        for i, s_i in enumerate(set_of):
            assert out.records[i].tobytes() == res.records[s_i].tobytes()
            for w in range(n_w):
                assert out.u_list(i, w).tolist() == res.u_list(s_i, w).tolist()
                assert out.q_list(i, w).tolist() == res.q_list(s_i, w).tolist()


def test_worker_pools_survive_a_fork():
    """The narrowing pass and the text writer run on process-wide worker pools; a fork()ed child inherits
    the pool object without its threads and must fall back to the calling thread instead of waiting for
    workers that do not exist (bench.py forks a multiprocessing pool after the library is loaded)."""
    import os
    import time

    from sai_amd.engine import to_int8_dosage

    g = np.random.default_rng(0).integers(-2, 3, (3000, 400)).astype(np.int64)
    want = to_int8_dosage(g)  # the pool exists in this process from here on
    pid = os.fork()
    if pid == 0:
        ok = np.array_equal(to_int8_dosage(g), want)
        os._exit(0 if ok else 1)
    for _ in range(200):  # 20 s: a hang is a failure, not a stuck suite
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            break
        time.sleep(0.1)
    else:
        os.kill(pid, 9)
        os.waitpid(pid, 0)
        raise AssertionError("the forked child hung in sai_narrow_to_int8")
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0


def test_to_int8_dosage():
    from sai_amd.engine import to_int8_dosage

    g = np.array([[0, 2, -2], [-300, 127, 1]], dtype=np.int64)
    assert to_int8_dosage(g).tolist() == [[0, 2, -2], [-128, 127, 1]]
    assert to_int8_dosage(g.astype(np.int8, casting="unsafe")[:1]).dtype == np.int8
    with pytest.raises(ValueError, match="above 127"):
        to_int8_dosage(np.array([[128]]))
    with pytest.raises(TypeError):
        to_int8_dosage(np.array([[0.5]]))
    with pytest.raises(ValueError, match="2-D"):
        to_int8_dosage(np.array([1, 2]))
    # the native pass against numpy on every integer dtype, contiguous and row-strided, and the
    # numpy fallback for column-strided views
    rng = np.random.default_rng(3)
    for dt in (np.int64, np.int32, np.int16, np.uint8, np.uint16, np.uint32, np.uint64):
        lo = max(np.iinfo(dt).min, -400)
        m = rng.integers(lo, 128, size=(257, 131)).astype(dt)
        want = np.maximum(m.astype(np.int64), -128).astype(np.int8)
        assert np.array_equal(to_int8_dosage(m), want)
        assert np.array_equal(to_int8_dosage(m[::2, 3:77]), want[::2, 3:77])      # rows strided, elements contiguous
        assert np.array_equal(to_int8_dosage(m[:, ::2]), want[:, ::2])           # elements strided: numpy path
        assert np.array_equal(to_int8_dosage(np.asfortranarray(m)), want)
        bad = m.copy()
        bad[200, 100] = 128
        with pytest.raises(ValueError, match="above 127"):
            to_int8_dosage(bad)
    assert to_int8_dosage(np.zeros((0, 5), dtype=np.int64)).shape == (0, 5)


# ---- outlier --------------------------------------------------------------


def test_outlier_equals_reference(tmp_path):
    """`sai outlier` output files and warnings, byte for byte, against the capture of the
    reference's outlier() (tests/golden/outlier.json; includes tests/data/test.q.scores)."""
    import warnings

    from sai_amd.sai import outlier

    g = load_golden("outlier.json")
    for run in g["runs"]:
        d = tmp_path / f"{run['table']}_{run['quantile']}"
        d.mkdir()
        (d / "scores.tsv").write_text(g["tables"][run["table"]])
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            outlier(score_file=str(d / "scores.tsv"), output_prefix=str(d / "o"), quantile=run["quantile"])
        files = {f.name[2:]: f.read_text() for f in sorted(d.iterdir()) if f.name.startswith("o.")}
        assert files == run["files"]
        assert [str(w.message) for w in wl if w.category is UserWarning] == run["warnings"]


def test_outlier_reference_pins_and_natural_order(in_repo_root, tmp_path):
    # reference tests/test_sai.py:154-173 and tests/utils/test_utils.py:450-511
    import pandas as pd

    from sai_amd.__main__ import main
    from sai_amd.utils import natsorted_df

    (tmp_path / "q.scores").write_text(load_golden("outlier.json")["tables"]["test.q.scores"])
    main(["outlier", "--score", str(tmp_path / "q.scores"), "--output-prefix", str(tmp_path / "outliers"), "--quantile", "0.25"])
    df = pd.read_csv(tmp_path / "outliers.Q.0.25.outliers.tsv", sep="\t")
    assert df["Q"].iloc[0] == 0.7
    df = pd.DataFrame({"Chrom": ["1", "10", "2", "X", "1"], "Start": [300, 50, 150, 10, 100], "End": [400, 100, 200, 50, 200]})
    s = natsorted_df(df)
    assert s["Chrom"].tolist() == ["1", "1", "2", "10", "X"] and s["Start"].tolist() == [100, 300, 150, 50, 10]
    with pytest.raises(ValueError, match="Missing required columns: End"):
        natsorted_df(pd.DataFrame({"Chrom": ["1"], "Start": [1]}))
    assert natsorted_df(pd.DataFrame(columns=["Chrom", "Start", "End"])).empty
    t = natsorted_df(pd.DataFrame({"Chrom": ["1", "2", "X"], "Start": ["100", "200", "300"], "End": ["150", "250", "350"]}))
    assert t["Start"].dtype == int and t["End"].dtype == int


def test_window_batch_bytes_round_trip_and_single_chunk_order(tmp_path):
    """The numeric batch a rank sends to rank 0 (records + CSR lists + optional f64 blocks) survives
    to_bytes / from_bytes bit for bit, and items_from_batches emits several chunks in the order ONE
    chunk would: combination-major, windows in chunk order."""
    from sai_amd.configs import StatConfig
    from sai_amd.engine import RECORD_DTYPE, WindowResults
    from sai_amd.preprocessors import FeaturePreprocessor
    from sai_amd.preprocessors.window_batch import ComboBatch, WindowBatch

    rng = np.random.default_rng(12)

    def combo(ref, tgt, w0, n_w, with_extra):
        rec = np.zeros((2, n_w), dtype=RECORD_DTYPE)
        rec["n_sites"] = rng.integers(0, 30, (1, n_w))
        rec["u_count"][0] = rng.integers(0, 4, n_w)
        rec["n_cond"][1] = rng.integers(0, 3, n_w)
        rec["n_cdd_q"][1] = np.minimum(rec["n_cond"][1], rng.integers(1, 3, n_w))
        rec["q"] = rng.random((2, n_w))
        rec["q"][1][rec["n_cond"][1] == 0] = np.nan
        off = np.zeros((2, n_w, 2), dtype=np.int64)
        for k, name in enumerate(("u_count", "n_cdd_q")):
            flat = rec[name].reshape(-1).astype(np.int64)
            off[:, :, k] = (np.cumsum(flat) - flat).reshape(2, n_w)
        uq = WindowResults(rec, off, rng.integers(1, 9999, int(rec["u_count"].sum())).astype(np.int32),
                           rng.integers(1, 9999, int(rec["n_cdd_q"].sum())).astype(np.int32))  # fmt: skip
        win = np.array([[1 + 100 * (w0 + i), 200 + 100 * (w0 + i)] for i in range(n_w)], dtype=np.int64)
        return ComboBatch(ref, tgt, ("s1", "s2"), None, win, rec[0]["n_sites"].astype(np.int32), ["U", "Q"], uq,
                          rng.random((n_w, 2, 4)) if with_extra else None, rng.random((n_w, 2)) if with_extra else None)  # fmt: skip

    chunks = [WindowBatch("9", [combo("r", "t1", 0, 3, True), combo("r", "t2", 0, 3, False)]),
              WindowBatch("9", [combo("r", "t1", 3, 2, True), combo("r", "t2", 3, 2, False)])]  # fmt: skip
    for b in chunks:
        back = WindowBatch.from_bytes(b.to_bytes())
        assert back.chr_name == b.chr_name and len(back.combos) == len(b.combos)
        for x, y in zip(back.combos, b.combos):
            assert (x.ref_pop, x.tgt_pop, x.src_comb, x.out_pop, x.uq_names, x.pos_dtype) == (y.ref_pop, y.tgt_pop, y.src_comb, y.out_pop, y.uq_names, y.pos_dtype)
            assert x.windows.tobytes() == y.windows.tobytes() and x.nsnps.tobytes() == y.nsnps.tobytes()
            assert x.uq.records.tobytes() == y.uq.records.tobytes() and np.array_equal(x.uq.offsets, y.uq.offsets)
            assert x.uq.cdd_u.tobytes() == y.uq.cdd_u.tobytes() and x.uq.cdd_q.tobytes() == y.uq.cdd_q.tobytes()
            for a, c in ((x.four, y.four), (x.dd, y.dd)):
                assert (a is None and c is None) or a.tobytes() == c.tobytes()
    with pytest.raises(ValueError):
        WindowBatch.from_bytes(b"nonsense")
    sc = StatConfig({"U": {"ref": {"r": 0.1}, "tgt": {"t1": 0.5, "t2": 0.5}, "src": {"s1": "=1", "s2": "=1"}},
                     "Q": {"ref": {"r": 0.1}, "tgt": {"t1": 0.9, "t2": 0.9}, "src": {"s1": "=1", "s2": "=1"}}})  # fmt: skip
    fp = FeaturePreprocessor(str(tmp_path / "o.tsv"), sc)
    merged = fp.items_from_batches([WindowBatch.from_bytes(b.to_bytes()) for b in chunks])
    assert [(it["tgt_pop"], it["start"]) for it in merged] == [("t1", 1 + 100 * i) for i in range(5)] + [("t2", 1 + 100 * i) for i in range(5)]
    per_chunk = [it for b in chunks for it in fp.items_from_batch(b)]
    assert sorted((it["tgt_pop"], it["start"], it["U"]) for it in merged) == sorted((it["tgt_pop"], it["start"], it["U"]) for it in per_chunk)
    for it in merged:
        assert isinstance(it["U"], int) and len(it["cdd_pos"]["U"]) == it["U"] or it["nsnps"] == 0


def test_value_radix_digits_are_an_exact_monotone_code():
    """The arithmetic of the windows stage's radix select on the VALUE (sai_amd/csrc/windows.hip,
    digit_on_path): digit l of v in [0, 1] is floor(v * 256^(l+1)) - 256 * floor(v * 256^l) -- scalings by
    powers of two and floors, so every step is exact in binary floating point -- and a value is on the
    chosen path iff floor(v * 256^l) equals the prefix the earlier digits spell.  Restated in numpy float64
    (the same IEEE operations) and driven like the kernel: histogram per level, the bin of the wanted rank,
    descend until the bin holds few values, rank those directly.  The selected order statistic must be
    np.sort's for tie-heavy inputs, neighbours one ulp apart, exact 0.0 / 1.0, subnormal-small values, and
    fifteen levels must separate any two distinct doubles."""
    rng = np.random.default_rng(2024)

    def digit(v, prefix, scale):
        s = v * scale
        on = np.ones(v.shape, bool) if scale == 1.0 else np.floor(s) == prefix
        d = np.floor(s * 256.0) - prefix * 256.0
        return np.where(on, d, -1.0).astype(np.int64)

    def select(vals, k, small=4):
        prefix, scale = 0.0, 1.0
        for level in range(15):
            d = digit(vals, prefix, scale)
            assert d.max() <= (256 if level == 0 else 255) and (d[d >= 0] >= 0).all()
            hist = np.bincount(d[d >= 0], minlength=257)
            before = np.concatenate([[0], np.cumsum(hist)])
            b = int(np.searchsorted(before, k, side="right") - 1)
            k -= int(before[b])
            members = vals[d == b]
            if len(members) <= small or level == 14:
                if len(members) > small:
                    assert np.all(members == members[0])  # the last level leaves only equal numbers in a bin
                    return members[0]
                return np.sort(members)[k]
            prefix, scale = prefix * 256.0 + b, scale * 256.0
        raise AssertionError("not reached")

    base = rng.random(300)
    ulp = np.nextafter(base[:40], 2.0)
    cases = [
        np.concatenate([base, ulp, np.nextafter(ulp, 2.0)]),  # neighbours one and two ulps apart
        np.concatenate([np.zeros(200), np.ones(150), rng.integers(0, 2001, 400) / 2000.0]),  # frequencies k / 2000 with heavy ties
        np.concatenate([rng.integers(0, 1398, 500) / 1398.0, [5e-324, 1e-310, 2.0**-55, 1.0 - 2.0**-53, 1.0]]),
        np.full(100, 1.0 / 3.0),
        np.array([0.0, 1.0]),
    ]
    for vals in cases:
        order = np.sort(vals)
        for k in sorted({0, 1, len(vals) // 3, len(vals) // 2, int(0.95 * (len(vals) - 1)), len(vals) - 2, len(vals) - 1} & set(range(len(vals)))):
            got = select(vals, k)
            assert got.tobytes() == order[k].tobytes(), (k, got, order[k])
    # the code is monotone: comparing digit strings compares the values
    v = np.sort(np.concatenate([rng.random(2000), ulp, [0.0, 1.0]]))
    codes = []
    for x in v:
        prefix, scale, ds = 0.0, 1.0, []
        for level in range(15):
            d = int(digit(np.array([x]), prefix, scale)[0])
            ds.append(d)
            prefix, scale = prefix * 256.0 + d, scale * 256.0
        codes.append(tuple(ds))
    assert codes == sorted(codes) and len(set(codes)) == len(set(v.tolist()))


def test_per_window_route_says_once_what_it_costs(monkeypatch):
    """After 64 one-window statistic calls in a process ONE warning names the batched route;
    the counter is the module's, so the classes and FeaturePreprocessor.run share it."""
    import warnings

    from sai_amd.stats import _window

    monkeypatch.setattr(_window, "_calls", 0)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        for _ in range(3 * _window.PER_WINDOW_WARN_AFTER):
            _window.note_per_window_call()
    mine = [w for w in seen if issubclass(w.category, _window.PerWindowRouteWarning)]
    assert len(mine) == 1
    text = str(mine[0].message)
    assert "ChunkPreprocessor.run" in text and "score_windows" in text and "PCIe" in text


def test_scan_while_reading_lets_device_failures_through(monkeypatch, tmp_path):
    """`score`'s preload under the chromosome scan (sai.py): a file-level error is repeated later by the usual
    read and fails there with its own message; a failure of the device is raised by the call that met it, also
    when the reader has wrapped it in the reference's ValueError (VERDICT r4 #8c)."""
    from sai_amd import _ffi, sai

    monkeypatch.setattr("sai_amd.utils.native_vcf.scan_first_last", lambda vcf, chrom: (10, 20))

    class Driver:
        def __init__(self, exc):
            self.exc = exc

        def preload(self, chr_name):
            raise self.exc

    assert sai._scan_while_reading(Driver(ValueError("Failed to read VCF file x from 1: bad genotype")), "x.vcf", "1") == ((10, 20), None)
    hip = _ffi.SaiHipError(_ffi.SAI_ERR_HIP, "hipMemcpyAsync failed: an illegal memory access was encountered")
    wrapped = ValueError("Failed to read VCF file x from 1: libsaihip error -2")
    wrapped.__cause__ = hip
    for exc in (hip, wrapped, MemoryError("out of memory")):
        with pytest.raises(type(exc)):
            sai._scan_while_reading(Driver(exc), "x.vcf", "1")
    refused = _ffi.SaiHipError(_ffi.SAI_ERR_UNSUPPORTED, "dosage outside int8")  # the library's own refusal: a file-level error
    assert sai._scan_while_reading(Driver(refused), "x.vcf", "1") == ((10, 20), None)
