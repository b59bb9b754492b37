"""GPU tests of the drop-in surface above the kernels: FeaturePreprocessor / ChunkPreprocessor /
score / CLI reproduce the reference's items and output text (golden vectors captured from the
reference, plus the pins of its own tests)."""

import json

import numpy as np
import pytest

from conftest import load_golden, same_f64, unhex
from test_host_logic import PIPE, _from_scenario

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import sai_amd.stats  # noqa: F401


@pytest.mark.parametrize("sc", PIPE, ids=[s["name"] for s in PIPE])
def test_run_windows_equals_reference_items_and_text(sc, tmp_path):
    from sai_amd.configs import PloidyConfig, StatConfig
    from sai_amd.preprocessors import FeaturePreprocessor
    from sai_amd.sai import write_headers

    wg, _ = _from_scenario(sc)
    stat_config = StatConfig(json.loads(json.dumps(sc["stats"])))
    out = tmp_path / "o.tsv"
    fp = FeaturePreprocessor(str(out), stat_config, sc["anc_allele_available"])
    items = fp.run_windows(wg)
    assert len(items) == len(sc["windows"])
    names = list(sc["stats"].keys())
    for it, win, exp in zip(items, sc["windows"], sc["items"]):
        assert [it["ref_pop"], it["tgt_pop"], list(it["src_pop_list"]), it["start"], it["end"], it["nsnps"]] == win[:6]
        assert it["out_pop"] == "NA"
        for k in names:
            if isinstance(exp[k], int):
                assert isinstance(it[k], int) and it[k] == exp[k]
            else:
                assert same_f64(it[k], unhex(exp[k]))
            assert np.asarray(it["cdd_pos"][k]).astype(np.int64).tolist() == exp[f"{k}_cdd"]
    write_headers(str(out), stat_config, PloidyConfig(sc["ploidies"]))
    fp.process_items(items)
    head = "Chrom\tStart\tEnd\tRef\tTgt\tSrc\tOutgroup\tN(Variants)\t" + "\t".join(names) + "\n"
    assert out.read_text() == head + sc["text"]["tsv"]
    for k in names:
        assert (tmp_path / f"o.{k}.log").read_text() == f"Chrom\tStart\tEnd\t{k}_SNP\n" + sc["text"][k]

    # the per-window plugin path (STAT_REGISTRY classes) gives the same items
    single = []
    for w in wg.get():
        single.extend(fp.run(**w))
    assert len(single) == len(items)
    for a, b in zip(single, items):
        assert a["nsnps"] == b["nsnps"]
        for k in names:
            assert same_f64(a[k], b[k]) if not isinstance(b[k], int) else a[k] == b[k]
            assert np.asarray(a["cdd_pos"][k]).tolist() == np.asarray(b["cdd_pos"][k]).tolist()


def test_feature_preprocessor_inline(tmp_path):
    # reference tests/preprocessors/test_feature_preprocessor.py:54-125
    from sai_amd.configs import PloidyConfig, StatConfig
    from sai_amd.preprocessors import FeaturePreprocessor

    g = load_golden("feature_inline.json")
    A = np.array
    sc = StatConfig({"DD": False,
                     "U": {"ref": {"ref1": 0.3}, "tgt": {"tgt1": 0.5}, "src": {"src1": "=1", "src2": "=1"}},
                     "Q": {"ref": {"ref1": 0.3}, "tgt": {"tgt1": 0.95}, "src": {"src1": "=0.2", "src2": "=0.4"}}})  # fmt: skip
    fp = FeaturePreprocessor(str(tmp_path / "t.tsv"), sc)
    kw = dict(chr_name="21", ref_pop="ref1", tgt_pop="tgt1", src_pop_list=["src1", "src2"], out_pop=None, start=1000,
              end=2000, pos=A([100, 200, 300]), ref_gts=A([[0, 0, 1], [1, 1, 0], [0, 1, 1]]),
              tgt_gts=A([[0, 1, 1], [1, 1, 1], [0, 0, 1]]),
              src_gts_list=[A([[0, 0, 0], [1, 0, 0], [1, 1, 1]]), A([[1, 1, 1], [0, 1, 1], [0, 0, 1]])], out_gts=None,
              ploidy_config=PloidyConfig({"ref": {"ref1": 1}, "tgt": {"tgt1": 1}, "src": {"src1": 1}}))  # fmt: skip
    full = fp.run(**kw)[0]
    assert full["U"] == g["full"]["U"] and same_f64(full["Q"], unhex(g["full"]["Q"]))
    assert full["nsnps"] == 3 and full["out_pop"] == "NA" and "DD" not in full
    assert full["src_pop_list"] == ["src1", "src2"] and (full["start"], full["end"]) == (1000, 2000)
    none = fp.run(**dict(kw, ref_gts=None, tgt_gts=None, src_gts_list=None, ploidy_config=None))[0]
    assert np.isnan(none["U"]) and np.isnan(none["Q"]) and none["cdd_pos"]["Q"].size == 0


def test_score_example_vcf(in_repo_root, tmp_path):
    """tests/test_sai.py:45-63 (Q == 0.9), test_feature_preprocessor.py:223 (U == 3),
    test_chunk_preprocessor.py:51-79; full text from the golden capture."""
    import yaml

    from sai_amd.configs import PloidyConfig, StatConfig
    from sai_amd.preprocessors import ChunkPreprocessor
    from sai_amd.sai import score

    ex = load_golden("example_vcf.json")
    out = tmp_path / "res" / "output.tsv"
    score(vcf_file="tests/data/example.vcf", chr_name="21", win_len=6666, win_step=6666, anc_allele_file=None,
          output_file=str(out), config="tests/data/test_sai.config.yaml", num_workers=1)  # fmt: skip
    assert out.read_text() == "Chrom\tStart\tEnd\tRef\tTgt\tSrc\tOutgroup\tN(Variants)\tQ\n" + ex["q_only"]["text"]["tsv"]
    assert (tmp_path / "res" / "output.Q.log").read_text() == "Chrom\tStart\tEnd\tQ_SNP\n" + ex["q_only"]["text"]["Q"]
    assert not (tmp_path / "res" / "output.U.log").exists()

    out2 = tmp_path / "uq.tsv"
    score(vcf_file="tests/data/example.vcf", chr_name="21", win_len=6666, win_step=6666, anc_allele_file=None,
          output_file=str(out2), config="tests/data/example.u_and_q.config.yaml", num_workers=1)  # fmt: skip
    assert out2.read_text().splitlines()[1] == "21\t1\t6666\tAFR\tCHB\tNean\tNA\t15\t3\t0.9"
    assert (tmp_path / "uq.U.log").read_text().splitlines()[1] == ex["u_and_q"]["text"]["U"].rstrip("\n")

    cfg = yaml.safe_load(open("tests/data/example.config.yaml"))
    pre = ChunkPreprocessor(
        vcf_file="tests/data/example.vcf", ref_ind_file="tests/data/example.ref.ind.list",
        tgt_ind_file="tests/data/example.tgt.ind.list", src_ind_file="tests/data/example.src.ind.list", out_ind_file=None,
        win_len=6666, win_step=6666, num_src=1, anc_allele_file=None, output_file=str(tmp_path / "c.tsv"),
        stat_config=StatConfig(cfg["statistics"]), ploidy_config=PloidyConfig(cfg["ploidies"]),
    )  # fmt: skip
    res = pre.run(chr_name="21", start=0, end=6666)
    assert res[0]["Q"] == 0.9 and res[0]["U"] == 1 and res[0]["cdd_pos"]["U"].tolist() == [777]


def test_score_mixed_ploidy_with_anc_alleles(in_repo_root, tmp_path):
    """tests/test_sai.py:127-151: gz VCF, tetraploid targets/sources, two sources, polarised:
    U of the two rows = 0 and 1 (the df columns of that test are outside this path)."""
    from sai_amd.sai import score

    out = tmp_path / "m.tsv"
    score(vcf_file="tests/data/test.mixed.ploidy.data.vcf.gz", chr_name="21", win_len=50000, win_step=50000,
          anc_allele_file="tests/data/test.mixed.ploidy.data.anc.alleles", output_file=str(out),
          config="tests/data/test_mixed_ploidy.u_only.config.yaml", num_workers=1)  # fmt: skip
    rows = [ln.split("\t") for ln in out.read_text().splitlines()]
    assert rows[0] == ["Chrom", "Start", "End", "Ref", "Tgt", "Src", "Outgroup", "N(Variants)", "U"]
    assert [r[8] for r in rows[1:]] == ["0", "1"] and rows[1][5] == "src1,src2"
    with pytest.raises(ValueError, match="outside the U/Q path"):
        score(vcf_file="tests/data/test.mixed.ploidy.data.vcf.gz", chr_name="21", win_len=50000, win_step=50000,
              anc_allele_file="tests/data/test.mixed.ploidy.data.anc.alleles", output_file=str(out),
              config="tests/data/test_mixed_ploidy.config.yaml", num_workers=1)  # fmt: skip


def test_cli_main(in_repo_root, tmp_path):
    from sai_amd.__main__ import main

    out = tmp_path / "cli.tsv"
    main(["score", "--vcf", "tests/data/example.vcf", "--chr-name", "21", "--win-len", "6666", "--win-step", "6666",
          "--output", str(out), "--config", "tests/data/test_sai.config.yaml"])  # fmt: skip
    assert out.read_text().splitlines()[1] == "21\t1\t6666\tAFR\tCHB\tNean\tNA\t15\t0.9"
    main(["score", "--vcf", "tests/data/test.data.vcf", "--chr-name", "21", "--win-len", "10000", "--win-step", "5000",
          "--anc-alleles", "tests/data/test.anc.allele.bed", "--output", str(out), "--config",
          "tests/data/test_mixed_ploidy.u_only.config.yaml"])  # fmt: skip
    assert len(out.read_text().splitlines()) == 1 + 2 * 10  # 10 windows x 2 target populations
