"""The native VCF tokenizer (libsaihip: sai_vcf_scan / sai_vcf_load) against the Python statement
of the same rules (sai_amd/utils/vcf.py), which is itself pinned to the reference tests'
expectations in test_host_logic.py.  Host-side only."""

import gzip

import numpy as np
import pytest


def write_vcf(path, rng, n_sites, n_samples, chroms=("7", "21", "22"), gz=False, crlf=False):
    """A deliberately awkward VCF: GT not always first in FORMAT, missing and half-missing calls,
    '/' and '|' separators, multi-allelic ALT, allele index 2, mixed ploidy per line, ragged
    allele counts, decoy chromosomes."""
    names = [f"s{i}" for i in range(n_samples)]
    lines = ["##fileformat=VCFv4.2", "##source=test", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names)]
    for chrom in chroms:
        pos = 0
        for _ in range(n_sites):
            pos += int(rng.integers(1, 90))
            ref = "ACGT"[int(rng.integers(4))]
            alt = "ACGT"[int(rng.integers(4))] + ("," + "ACGT"[int(rng.integers(4))] if rng.random() < 0.1 else "")
            fmt = str(rng.choice(["GT", "GT:DP", "DP:GT", "DP:GQ:GT:PL"]))
            gi = fmt.split(":").index("GT")
            calls = []
            for _ in range(n_samples):
                pl = int(rng.choice([1, 2, 2, 2, 4]))
                alle = [str(rng.choice([".", "0", "0", "0", "1", "1", "2"])) for _ in range(pl)]
                gt = str(rng.choice(["|", "/"])).join(alle)
                sub = [str(int(rng.integers(0, 99))) for _ in fmt.split(":")]
                sub[gi] = gt
                calls.append(":".join(sub))
            lines.append("\t".join([chrom, str(pos), ".", ref, alt, "100", "PASS", "AA=" + ref, fmt] + calls))
    text = ("\r\n" if crlf else "\n").join(lines) + ("\r\n" if crlf else "\n")
    if gz == "bgzf":
        write_bgzf(path, text.encode(), rng)
    elif gz:
        with gzip.open(path, "wt", newline="") as f:
            f.write(text)
    else:
        with open(path, "w", newline="") as f:
            f.write(text)
    return names


def write_bgzf(path, data: bytes, rng, max_block=6000):
    """bgzip container (SAM spec 4.1): independent gzip members with a 'BC' extra subfield holding
    the member size - 1, ragged small blocks so that lines straddle members, empty EOF member."""
    import struct
    import zlib

    def member(chunk: bytes) -> bytes:
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = comp.compress(chunk) + comp.flush()
        bsize = 12 + 6 + len(raw) + 8
        head = b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        return head + raw + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))

    with open(path, "wb") as f:
        off = 0
        while off < len(data):
            n = int(rng.integers(1, max_block))
            f.write(member(data[off : off + n]))
            off += n
        f.write(member(b""))


def python_reader(path, chrom, names, ploidy, start=None, end=None, anc=None):
    from sai_amd.utils.read_data import _load_python

    return _load_python(str(path), chrom, list(names), ploidy, start, end, anc)


@pytest.mark.parametrize("gz,crlf", [(False, False), (True, False), (False, True), ("bgzf", False), ("bgzf", True)])
def test_native_equals_python_reader(tmp_path, gz, crlf):
    from sai_amd.utils.native_vcf import load_dosage, scan_first_last
    from sai_amd.utils.vcf import first_last_pos

    rng = np.random.default_rng(11 + bool(gz) + 2 * crlf + 4 * (gz == "bgzf"))
    path = tmp_path / ("t.vcf.gz" if gz else "t.vcf")
    names = write_vcf(path, rng, 400, 13, gz=gz, crlf=crlf)
    bed = tmp_path / "anc.bed"
    with open(bed, "w") as f:
        # ancestral alleles for ~70 % of chr21 sites: REF, ALT, or neither
        from sai_amd.utils.vcf import read_region

        reg = read_region(str(path), "21", names[:1])
        for p, r, a in zip(reg.pos, reg.ref, reg.alt):
            u = rng.random()
            if u < 0.3:
                continue
            allele = r if u < 0.6 else (a if u < 0.9 else "-")
            f.write(f"21\t{p - 1}\t{p}\t{allele}\n")
        f.write("22\t9\t10\tA\n")
    assert scan_first_last(str(path), "21") == first_last_pos(str(path), "21")
    assert scan_first_last(str(path), "nope") == (None, None)
    pick = [names[i] for i in (5, 0, 12, 3, 7)]
    for ploidy in (1, 2, 3, 4):
        for start, end in ((None, None), (500, 9000), (9001, 9001), (10**7, None)):
            for anc in (None, str(bed)):
                for threads in (1, 5):
                    pos, dos, n_matched, n_anc = load_dosage(str(path), "21", pick, [ploidy] * len(pick), start, end, anc, threads)
                    epos, edos, ematched, eanc = python_reader(path, "21", pick, ploidy, start, end, anc)
                    assert pos.tolist() == epos.tolist() and pos.dtype == np.int32
                    assert dos.dtype == np.int8 and np.array_equal(dos, edos)
                    assert n_matched == ematched
                    if anc and ematched:
                        assert n_anc == eanc
    # mixed ploidy per sample in one call
    pos, dos, _, _ = load_dosage(str(path), "7", names[:4], [1, 2, 3, 4])
    for j, pl in enumerate([1, 2, 3, 4]):
        epos, edos, _, _ = python_reader(path, "7", names[j : j + 1], pl)
        assert np.array_equal(dos[:, j], edos[:, 0]) and pos.tolist() == epos.tolist()


def test_native_on_reference_fixtures(in_repo_root):
    """The reference's own test VCFs through both engines of read_data."""
    from sai_amd.configs import PloidyConfig
    from sai_amd.utils import read_data

    cases = [
        ("tests/data/example.vcf", "21", {"ref": {"AFR": 2}, "tgt": {"CHB": 2}, "src": {"Nean": 2}},
         ("tests/data/example.ref.ind.list", "tests/data/example.tgt.ind.list", "tests/data/example.src.ind.list"), None),
        ("tests/data/test.data.vcf", "21", {"ref": {"ref1": 2}, "tgt": {"tgt1": 2, "tgt2": 2}, "src": {"src1": 2, "src2": 2}},
         ("tests/data/test.ref.ind.list", "tests/data/test.tgt.ind.list", "tests/data/test.src.ind.list"), "tests/data/test.anc.allele.bed"),
        ("tests/data/test.mixed.ploidy.data.vcf.gz", "21", {"ref": {"ref1": 2}, "tgt": {"tgt1": 4, "tgt2": 4}, "src": {"src1": 4, "src2": 4}},
         ("tests/data/test.ref.ind.list", "tests/data/test.tgt.ind.list", "tests/data/test.src.ind.list"), "tests/data/test.mixed.ploidy.data.anc.alleles"),
    ]  # fmt: skip
    for vcf, chrom, pl, inds, anc in cases:
        for region in ((None, None), (2000, 30000)):
            a = read_data(vcf, chrom, PloidyConfig(pl), *inds, anc_allele_file=anc, start=region[0], end=region[1])
            b = read_data(vcf, chrom, PloidyConfig(pl), *inds, anc_allele_file=anc, start=region[0], end=region[1], engine="python")
            for g in ("ref", "tgt", "src"):
                assert a[g][1] == b[g][1]
                assert (a[g][0] is None) == (b[g][0] is None)
                for pop in a[g][0] or {}:
                    assert a[g][0][pop].POS.tolist() == b[g][0][pop].POS.tolist()
                    assert np.array_equal(a[g][0][pop].GT, b[g][0][pop].GT) and a[g][0][pop].GT.dtype == np.int8


def test_native_errors(tmp_path, in_repo_root):
    from sai_amd.configs import PloidyConfig
    from sai_amd.utils import read_data
    from sai_amd.utils.native_vcf import load_dosage, scan_first_last

    with pytest.raises(ValueError, match="samples not found"):
        load_dosage("tests/data/example.vcf", "21", ["nobody"], [2])
    with pytest.raises(ValueError, match="cannot open"):
        scan_first_last(str(tmp_path / "missing.vcf"), "21")
    bad = tmp_path / "bad.vcf"
    bad.write_text("21\t5\t.\tA\tT\t.\t.\t.\tGT\t0|0\n")
    with pytest.raises(ValueError, match="no #CHROM header"):
        load_dosage(str(bad), "21", ["x"], [2])
    weird = tmp_path / "weird.vcf"
    weird.write_text("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ta\n21\t5\t.\tA\tT\t.\t.\t.\tGT\t0|x\n")
    with pytest.raises(ValueError, match="unparsable genotype at 21:5"):
        load_dosage(str(weird), "21", ["a"], [2])
    pc = PloidyConfig({"ref": {"ref1": 2}, "tgt": {"tgt1": 2, "tgt2": 2}, "src": {"src1": 2, "src2": 2}})
    for engine in ("native", "python"):
        with pytest.raises(ValueError, match="No ancestral allele is found for chromosome 21 in the region 16000-20000"):
            read_data("tests/data/test.data.vcf", "21", pc, "tests/data/test.ref.ind.list", "tests/data/test.tgt.ind.list",
                      None, anc_allele_file="tests/data/test.anc.allele.bed", start=16000, end=20000, engine=engine)  # fmt: skip
        # a region without records never looks at the ancestral alleles (utils.py:143-144)
        empty = read_data("tests/data/test.data.vcf", "21", pc, "tests/data/test.ref.ind.list", "tests/data/test.tgt.ind.list",
                          None, anc_allele_file="tests/data/test.anc.allele.bed", start=100, end=2000, engine=engine)  # fmt: skip
        assert empty["ref"][0] is None and empty["tgt"][0] is None


def test_bgzf_batches_and_damage(tmp_path, monkeypatch):
    """Tiny batches (every member boundary is also a batch boundary with a carried partial line)
    give the same matrix as one batch; a flipped byte or a cut file is an error, not garbage."""
    from sai_amd.utils.native_vcf import load_dosage, scan_first_last

    rng = np.random.default_rng(77)
    path = tmp_path / "b.vcf.gz"
    names = write_vcf(path, rng, 300, 9, gz="bgzf")
    want = load_dosage(str(path), "21", names, [2] * len(names), None, None, None, 3)
    first_last = scan_first_last(str(path), "22")
    for batch in ("1", "700", "5000"):
        monkeypatch.setenv("SAI_VCF_BATCH_BYTES", batch)
        got = load_dosage(str(path), "21", names, [2] * len(names), None, None, None, 3)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2:] == want[2:]
        assert scan_first_last(str(path), "22") == first_last
    monkeypatch.delenv("SAI_VCF_BATCH_BYTES")
    raw = bytearray(path.read_bytes())
    cut = tmp_path / "cut.vcf.gz"
    cut.write_bytes(bytes(raw[: len(raw) // 2]))
    with pytest.raises((ValueError, OSError), match="BGZF"):
        load_dosage(str(cut), "21", names, [2] * len(names))
    raw[len(raw) // 2] ^= 0x5A
    bad = tmp_path / "bad.vcf.gz"
    bad.write_bytes(bytes(raw))
    with pytest.raises((ValueError, OSError), match="BGZF"):
        load_dosage(str(bad), "21", names, [2] * len(names))
