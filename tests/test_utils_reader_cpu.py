"""The allele-level reader surface of ``sai.utils`` (sai/utils/utils.py:78-232, 359-432,
540-555) against the expected arrays the reference's own tests publish (tests/golden/utils_reader.json),
against an independent parse of the raw VCF text, and against the native dosage tokenizer where the two
must agree."""

import numpy as np
import pytest

from conftest import load_golden

G = load_golden("utils_reader.json")


def block(case):
    from sai_amd.utils import ChromosomeData

    return ChromosomeData(POS=np.array(case["POS"]), REF=np.array(case["REF"]), ALT=np.array(case["ALT"]), GT=np.array(case["GT"], dtype=np.int8))


def test_filter_geno_data_keeps_the_indexed_rows_of_every_field():
    from sai_amd.utils import filter_geno_data

    case = G["filter_geno_data"]
    got = filter_geno_data(block(case), np.array(case["index"]))
    assert got.POS.tolist() == case["expect"]["POS"] and got.REF.tolist() == case["expect"]["REF"]
    assert got.ALT.tolist() == case["expect"]["ALT"] and list(got.GT.shape) == case["expect"]["GT_shape"]
    by_number = filter_geno_data(block(case), np.flatnonzero(case["index"]))  # "boolean or integer array"
    assert by_number.POS.tolist() == got.POS.tolist() and np.array_equal(by_number.GT, got.GT)
    from sai_amd.utils import ChromosomeData

    bare = ChromosomeData(POS=np.array(case["POS"]), REF=None, ALT=None, GT=np.array(case["GT"], dtype=np.int8).sum(axis=2))
    kept = filter_geno_data(bare, np.array(case["index"]))  # a dosage block of the native ingest: no REF / ALT to filter
    assert kept.REF is None and kept.ALT is None and kept.POS.tolist() == case["expect"]["POS"] and kept.GT.shape == (3, 2)


def test_filter_fixed_variants_drops_all_hom_ref_and_all_hom_alt_sites():
    from sai_amd.utils import filter_fixed_variants

    case = G["filter_fixed_variants"]
    got = filter_fixed_variants({p: block(case[p]) for p in ("pop1", "pop2")}, case["samples"])
    e1, e2 = case["expect"]["pop1"], case["expect"]["pop2"]
    assert got["pop1"].POS.tolist() == e1["POS"] and got["pop1"].REF.tolist() == e1["REF"] and got["pop1"].ALT.tolist() == e1["ALT"]
    assert list(got["pop1"].GT.shape) == e1["GT_shape"]
    assert got["pop2"].POS.size == got["pop2"].REF.size == got["pop2"].ALT.size == got["pop2"].GT.size == e2["GT_size"] == 0
    # the rule counts homozygous-alternate CALLS (utils.py:380-381): (1, 1) next to (2, 2) is "fixed", a (1, 2) call is not
    from sai_amd.utils import ChromosomeData

    mixed = ChromosomeData(POS=np.array([7]), REF=np.array(["A"]), ALT=np.array(["C"]), GT=np.array([[[1, 1], [2, 2]]], dtype=np.int8))
    assert filter_fixed_variants({"p": mixed}, {"p": ["a", "b"]})["p"].POS.tolist() == []
    half = ChromosomeData(POS=np.array([7]), REF=np.array(["A"]), ALT=np.array(["C"]), GT=np.array([[[1, 2], [1, 1]]], dtype=np.int8))
    assert filter_fixed_variants({"p": half}, {"p": ["a", "b"]})["p"].POS.tolist() == [7]


def test_flip_snps_mirrors_every_allele_of_the_listed_positions_in_place():
    from sai_amd.utils import flip_snps

    case = G["flip_snps"]
    data = block(case)
    flip_snps(data, case["flipped_snps"])
    assert data.GT.tolist() == case["expect_GT"]
    data.GT[2, 0] = [-1, 0]  # a missing allele of a flipped site becomes 2 (|-1 - 1|): the reference's behaviour
    flip_snps(data, [300])
    assert data.GT[2].tolist() == [[2, 1], [1, 0]]


def test_read_anc_allele_and_get_ref_alt_allele(in_repo_root):
    from sai_amd.utils import get_ref_alt_allele, read_anc_allele, read_geno_data

    case = G["read_anc_allele"]
    assert read_anc_allele(case["file"], case["chr_name"]) == {c: {int(p): a for p, a in t.items()} for c, t in case["expect"].items()}
    d = read_geno_data(vcf="tests/data/test.data.vcf", ind_samples={"tgt1": ["ind1", "ind2"]}, chr_name="21", filter_missing=False)
    ref, alt = get_ref_alt_allele(d["tgt1"].REF, d["tgt1"].ALT, d["tgt1"].POS)
    want = G["get_ref_alt_allele"]
    assert {str(p): a for p, a in ref.items()} == want["ref"] and {str(p): a for p, a in alt.items()} == want["alt"]


def raw_calls(path, chrom, names, ploidy):
    """Independent of the package: allele calls straight from the VCF text."""
    out, pos, ref, alt = [], [], [], []
    for line in open(path):
        if line.startswith("##"):
            continue
        f = line.rstrip("\n").split("\t")
        if line.startswith("#CHROM"):
            cols = [f.index(n) for n in names]
            continue
        if f[0] != chrom:
            continue
        gi = f[8].split(":").index("GT")
        row = []
        for c in cols:
            al = [(-1 if a == "." else int(a)) for a in f[c].split(":")[gi].replace("|", "/").split("/")][:ploidy]
            row.append(al + [-1] * (ploidy - len(al)))
        out.append(row)
        pos.append(int(f[1]))
        ref.append(f[3])
        alt.append(f[4].split(",")[0])
    return np.array(pos), ref, alt, np.array(out, dtype=np.int8)


def test_read_data_phased_blocks_are_the_files_allele_calls(in_repo_root):
    """utils.py:215-356 with the reference's defaults except the fixed-variant filters (its own test,
    test_utils.py:152-201): haplotype columns [sites][individuals * ploidy], REF / ALT kept."""
    from sai_amd.configs import PloidyConfig
    from sai_amd.utils import parse_ind_file, read_data

    pc = PloidyConfig({"ref": {"ref1": 2}, "tgt": {"tgt1": 2, "tgt2": 2}, "src": {"src1": 2, "src2": 2}})
    res = read_data(vcf_file="tests/data/test.data.vcf", chr_name="21", ref_ind_file="tests/data/test.ref.ind.list",
                    tgt_ind_file="tests/data/test.tgt.ind.list", src_ind_file=None, out_ind_file=None, anc_allele_file=None,
                    filter_ref=False, filter_tgt=False, filter_src=False, filter_out=False, ploidy_config=pc)  # fmt: skip
    assert res["ref"][1] == parse_ind_file("tests/data/test.ref.ind.list") and res["tgt"][1] == parse_ind_file("tests/data/test.tgt.ind.list")
    assert res["src"] == (None, None) and res["outgroup"] == (None, None)
    shape = G["read_data_phased_shape"]
    for group, pop, names in (("ref", "ref1", ["ind5", "ind6"]), ("tgt", "tgt2", ["ind3", "ind4"]), ("tgt", "tgt1", ["ind1", "ind2"])):
        pos, ref, alt, calls = raw_calls("tests/data/test.data.vcf", "21", names, 2)
        b = res[group][0][pop]
        assert b.GT.shape == (shape["n_sites"], shape["n_columns"]) and np.array_equal(b.GT, calls.reshape(len(pos), 4))
        assert b.POS.tolist() == pos.tolist() and b.REF.tolist() == ref and b.ALT.tolist() == alt


def test_read_data_with_ancestral_alleles_matches_the_references_published_blocks(in_repo_root):
    from sai_amd.configs import PloidyConfig
    from sai_amd.utils import read_data

    pc = PloidyConfig({"ref": {"ref1": 2}, "tgt": {"tgt1": 2, "tgt2": 2}, "src": {"src1": 2, "src2": 2}})
    data = read_data(vcf_file="tests/data/test.data.vcf", chr_name="21", ref_ind_file="tests/data/test.ref.ind.list",
                     tgt_ind_file="tests/data/test.tgt.ind.list", src_ind_file=None, out_ind_file=None,
                     anc_allele_file="tests/data/test.anc.allele.bed", filter_ref=False, filter_tgt=False, ploidy_config=pc)  # fmt: skip
    want = G["check_anc_allele_through_read_data"]
    assert np.array_equal(data["ref"][0]["ref1"].GT, np.array(want["ref1_GT"]).reshape(3, 4))
    assert np.array_equal(data["tgt"][0]["tgt1"].GT, np.array(want["tgt1_GT"]).reshape(3, 4))
    assert np.array_equal(data["tgt"][0]["tgt2"].GT, np.array(want["tgt2_GT"]).reshape(3, 4))
    assert data["tgt"][0]["tgt1"].POS.tolist() == want["tgt_pos"] == data["tgt"][0]["tgt2"].POS.tolist()


def test_the_options_compose_and_the_unfiltered_unphased_form_is_the_native_tokenizers(in_repo_root):
    """Unphased + no filters = what ``score`` asks for = the native dosage tokenizer; every other
    combination comes from the allele-level reader and must reduce to it: summing a phased block's
    haplotype columns gives the dosages, the filters only remove sites."""
    from sai_amd.configs import PloidyConfig
    from sai_amd.utils import read_data, read_dosage_data, read_geno_data

    pc = PloidyConfig({"ref": {"ref1": 2}, "tgt": {"tgt1": 4, "tgt2": 4}, "src": {"src1": 4, "src2": 4}})
    files = dict(vcf_file="tests/data/test.mixed.ploidy.data.vcf.gz", chr_name="21", ploidy_config=pc,
                 ref_ind_file="tests/data/test.ref.ind.list", tgt_ind_file="tests/data/test.tgt.ind.list",
                 src_ind_file="tests/data/test.src.ind.list", out_ind_file=None)  # fmt: skip
    off = dict(filter_ref=False, filter_tgt=False, filter_src=False, filter_out=False, filter_missing=False)
    for anc in (None, "tests/data/test.mixed.ploidy.data.anc.alleles"):
        fast = read_data(**files, anc_allele_file=anc, is_phased=False, **off)
        same = read_dosage_data(**files, anc_allele_file=anc)
        phased = read_data(**files, anc_allele_file=anc, is_phased=True, **off)
        for group in ("ref", "tgt", "src"):
            for pop, b in fast[group][0].items():
                assert np.array_equal(b.GT, same[group][0][pop].GT) and b.GT.dtype == np.int8
                pl = pc.root[group][pop]
                hap = phased[group][0][pop]
                assert hap.GT.shape == (len(b.POS), b.GT.shape[1] * pl) and hap.POS.tolist() == b.POS.tolist()
                assert np.array_equal(hap.GT.reshape(len(b.POS), -1, pl).sum(axis=2), b.GT)
        # filters: missing calls, then fixed variants -- subsets of the unfiltered sites, rule by rule
        filtered = read_data(**files, anc_allele_file=anc, is_phased=True, filter_ref=True, filter_tgt=True, filter_src=True,
                             filter_missing=True)  # fmt: skip
        for group in ("ref", "tgt", "src"):
            for pop, b in filtered[group][0].items():
                pl = pc.root[group][pop]
                full = phased[group][0][pop]
                calls = full.GT.reshape(len(full.POS), -1, pl)
                missing = (calls < 0).any(axis=(1, 2))
                hom_ref = (calls == 0).all(axis=2).all(axis=1)
                hom_alt = ((calls[:, :, :1] > 0) & (calls == calls[:, :, :1])).all(axis=2).all(axis=1)
                keep = ~missing & ~hom_ref & ~hom_alt
                assert b.POS.tolist() == full.POS[keep].tolist() and np.array_equal(b.GT, full.GT[keep])
    # read_geno_data: several populations from one read, None for an empty region, the reference's error text
    two = read_geno_data("tests/data/test.data.vcf", {"a": ["ind1"], "b": ["ind5", "ind6"]}, "21", filter_missing=False)
    assert two["a"].GT.shape == (19, 1, 2) and two["b"].GT.shape == (19, 2, 2) and two["a"].REF.dtype.kind == "U"
    assert read_geno_data("tests/data/test.data.vcf", {"a": ["ind1"]}, "21", start=3, end=9) is None
    with pytest.raises(ValueError, match="Failed to read VCF file nope.vcf from 21:1-5"):
        read_geno_data("nope.vcf", {"a": ["ind1"]}, "21", start=1, end=5)
    with pytest.raises(ValueError, match=r"Population 'tgt2' in ploidy_config\[tgt\] not found in sample file"):
        read_data(**{**files, "tgt_ind_file": "tests/data/test.ref.ind.list", "ploidy_config": PloidyConfig(
            {"ref": {"ref1": 2}, "tgt": {"ref1": 2, "tgt2": 2}, "src": {"src1": 4, "src2": 4}})}, anc_allele_file=None)
    with pytest.warns(RuntimeWarning, match="Population 'tgt2' found in sample file but not in ploidy_config"):
        only = read_data(**{**files, "ploidy_config": PloidyConfig({"ref": {"ref1": 2}, "tgt": {"tgt1": 4}, "src": {"src1": 4, "src2": 4}})},
                         anc_allele_file=None)  # fmt: skip
    assert list(only["tgt"][0]) == ["tgt1"]


@pytest.mark.parametrize("gz", [False, True, "bgzf"])
def test_allele_calls_from_the_native_tokenizer_equal_the_readable_reader(tmp_path, gz):
    """The allele-level reader takes its calls from the native tokenizer (allele k = dosage of the first k
    alleles minus the dosage of the first k - 1: one ``sai_vcf_load`` pass per allele) and only REF / ALT from
    the text.  On deliberately awkward files -- GT not first in FORMAT, missing and half-missing calls, '/' and
    '|', multi-allelic ALT and allele index 2, calls shorter and longer than the ploidy asked for, plain / gzip
    / bgzip, regions -- every call, position and allele string equals the package's Python statement of the
    rules, and so do the blocks of ``read_geno_data`` / ``read_data`` built on it."""
    from test_ingest_native import write_vcf

    from sai_amd.configs import PloidyConfig
    from sai_amd.utils import read_data, read_geno_data
    from sai_amd.utils.geno import allele_calls

    rng = np.random.default_rng(7)
    path = tmp_path / ("a.vcf.gz" if gz else "a.vcf")
    names = write_vcf(path, rng, 260, 14, gz=gz)
    pick = [names[i] for i in (9, 2, 13, 0, 5)]
    for chrom, region in (("7", (None, None)), ("21", (300, 9000)), ("22", (1, 40))):
        for ploidy in (1, 2, 3, 4):
            a = allele_calls(str(path), chrom, pick, ploidy, *region, engine="native")
            b = allele_calls(str(path), chrom, pick, ploidy, *region, engine="python")
            assert a[0].tolist() == b[0].tolist() and list(a[1]) == list(b[1]) and list(a[2]) == list(b[2])
            assert a[3].dtype == np.int8 and np.array_equal(a[3], b[3]), (chrom, region, ploidy)
    full = allele_calls(str(path), "7", pick, 4)[3]
    assert full.min() == -1 and full.max() == 2 and len(full) == 260  # missing alleles and a second alternate allele are in the data
    for missing in (True, False):
        x = read_geno_data(str(path), {"p": pick[:3], "q": pick[3:]}, "7", ploidy=2, filter_missing=missing)
        y = read_geno_data(str(path), {"p": pick[:3], "q": pick[3:]}, "7", ploidy=2, filter_missing=missing, engine="python")
        for pop in ("p", "q"):
            assert x[pop].POS.tolist() == y[pop].POS.tolist() and np.array_equal(x[pop].GT, y[pop].GT)
            assert x[pop].REF.tolist() == y[pop].REF.tolist() and x[pop].ALT.tolist() == y[pop].ALT.tolist()
    ind = tmp_path / "ind.list"
    ind.write_text("".join(f"A\t{n}\n" for n in pick[:3]) + "".join(f"B\t{n}\n" for n in pick[3:]))
    pc = PloidyConfig({"ref": {"A": 2}, "tgt": {"B": 4}, "src": {"A": 1}})
    kw = dict(vcf_file=str(path), chr_name="21", ploidy_config=pc, ref_ind_file=str(ind), tgt_ind_file=str(ind), src_ind_file=str(ind),
              out_ind_file=None, anc_allele_file=None, filter_missing=False)  # fmt: skip
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)  # populations of the list without a ploidy entry are skipped, as in the reference
        n_, p_ = read_data(**kw), read_data(**kw, engine="python")
    for group, pop, width in (("ref", "A", 6), ("tgt", "B", 8), ("src", "A", 3)):
        assert n_[group][0][pop].GT.shape[1] == width and np.array_equal(n_[group][0][pop].GT, p_[group][0][pop].GT)
