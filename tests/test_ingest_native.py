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
    from sai_amd.utils import read_dosage_data as read_data

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
    from sai_amd.utils import read_dosage_data as read_data
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


def write_tbi(gz_path, one_chunk=False):
    """A tabix index (TBI, SAM/tabix spec) for a bgzip VCF written by ``write_bgzf``: bins with one
    chunk each, the linear index with the htslib back-fill, plain-gzip compressed.  ``one_chunk``: every
    record of a chromosome in bin 0 -- one chunk from its first record to its last, as a coarse writer
    may leave it (legal: a bin's chunk only has to cover the bin's records)."""
    import struct
    import zlib

    raw = open(gz_path, "rb").read()
    members, off, upos = [], 0, 0  # (compressed offset, uncompressed offset, text)
    text = b""
    while off < len(raw):
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        chunk = zlib.decompress(raw[off + 18 : off + bsize - 8], -15)
        members.append((off, upos, len(chunk)))
        text += chunk
        upos += len(chunk)
        off += bsize

    def voff(t):
        for coff, u0, n in members:
            if u0 <= t < u0 + n:
                return (coff << 16) | (t - u0)
        return (len(raw) << 16)  # end of file

    def reg2bin(beg, end):
        end -= 1
        for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
            if beg >> shift == end >> shift:
                return base + (beg >> shift)
        return 0

    refs, order = {}, []
    t = 0
    for line in text.split(b"\n"):
        start_t, t = t, t + len(line) + 1
        if not line or line.startswith(b"#"):
            continue
        f = line.split(b"\t", 5)
        name, beg = f[0].decode(), int(f[1]) - 1
        end = beg + len(f[3])
        if name not in refs:
            refs[name] = {"bins": {}, "lin": {}}
            order.append(name)
        r = refs[name]
        v0, v1 = voff(start_t), voff(t)
        b = r["bins"].setdefault(0 if one_chunk else reg2bin(beg, end), [v0, v1])
        b[1] = v1
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            r["lin"][w] = min(r["lin"].get(w, v0), v0)
    names = b"".join(n.encode() + b"\0" for n in order)
    out = b"TBI\1" + struct.pack("<8i", len(order), 2, 1, 2, 0, ord("#"), 0, len(names)) + names
    for n in order:
        r = refs[n]
        out += struct.pack("<i", len(r["bins"]))
        for b, (v0, v1) in sorted(r["bins"].items()):
            out += struct.pack("<Ii", b, 1) + struct.pack("<QQ", v0, v1)
        n_intv = max(r["lin"]) + 1
        lin, prev = [], min(r["lin"].values())
        for w in range(n_intv):
            prev = r["lin"].get(w, prev)
            lin.append(prev)
        out += struct.pack("<i", n_intv) + struct.pack(f"<{n_intv}Q", *lin)
    with gzip.open(str(gz_path) + ".tbi", "wb") as f:
        f.write(out)
    return members


def test_tabix_index_seek_equals_full_pass(tmp_path, monkeypatch):
    """With <vcf>.tbi next to a bgzip VCF a region load seeks through the linear index and stops
    after the region, the chromosome scan reads two records: same results as without the index --
    and a damaged block outside the region is never touched (proof that it seeks)."""
    import os

    from sai_amd.utils.native_vcf import load_dosage, scan_first_last

    rng = np.random.default_rng(5)
    path = tmp_path / "i.vcf.gz"
    names = write_vcf(path, rng, 1500, 7, gz="bgzf")  # ~1500 * 45 bp: several 16 kb windows per chromosome
    regions = [(1, 10**9), (1, 1), (5000, 40000), (16384, 16385), (16385, 32768), (70000, 70010), (10**6, 10**7), (33000, None)]
    plain = {}
    for chrom in ("7", "21", "22", "nope"):
        plain[chrom, "scan"] = scan_first_last(str(path), chrom)
        for reg in regions:
            plain[chrom, reg] = load_dosage(str(path), chrom, names, [2] * len(names), reg[0], reg[1], None, 3)
    members = write_tbi(path)
    for chrom in ("7", "21", "22", "nope"):
        assert scan_first_last(str(path), chrom) == plain[chrom, "scan"]
        for reg in regions:
            got = load_dosage(str(path), chrom, names, [2] * len(names), reg[0], reg[1], None, 3)
            want = plain[chrom, reg]
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2], (chrom, reg)
    # damage a block in the middle of the file (inside chromosome 21): regions of 7 and 22 still load
    raw = bytearray(path.read_bytes())
    victim = members[len(members) // 2][0]
    raw[victim + 30] ^= 0xFF
    path.write_bytes(bytes(raw))
    os.utime(str(path) + ".tbi")  # the index stays newer than the rewritten file (a stale index is ignored)
    monkeypatch.setenv("SAI_VCF_BATCH_BYTES", "20000")  # the whole test file is smaller than a default batch
    for chrom, reg in (("7", (5000, 40000)), ("22", (5000, 40000))):
        got = load_dosage(str(path), chrom, names, [2] * len(names), reg[0], reg[1], None, 3)
        assert np.array_equal(got[1], plain[chrom, reg][1])
        assert scan_first_last(str(path), chrom) == plain[chrom, "scan"]  # two short reads, none of them near the damage
    raw[victim + 30] ^= 0xFF
    path.write_bytes(bytes(raw))  # intact again, to write the coarse index from it
    write_tbi(path, one_chunk=True)
    raw[victim + 30] ^= 0xFF
    path.write_bytes(bytes(raw))
    os.utime(str(path) + ".tbi")
    # one chunk per chromosome: the last record is found through the linear index's last entry -- also for
    # chromosome 21, whose chunk runs across the damaged member
    for chrom in ("7", "21", "22"):
        assert scan_first_last(str(path), chrom) == plain[chrom, "scan"], chrom
    raw[victim + 30] ^= 0xFF
    path.write_bytes(bytes(raw))
    write_tbi(path)
    raw[victim + 30] ^= 0xFF
    path.write_bytes(bytes(raw))
    os.utime(str(path) + ".tbi")
    # an index older than the file is not trusted: the full pass runs and meets the damaged block
    st = os.stat(path)
    os.utime(str(path) + ".tbi", (st.st_atime - 100, st.st_mtime - 100))
    with pytest.raises((ValueError, OSError), match="BGZF"):
        load_dosage(str(path), "22", names, [2] * len(names), 5000, 40000, None, 3)
    os.remove(str(path) + ".tbi")
    with pytest.raises((ValueError, OSError), match="BGZF"):
        load_dosage(str(path), "22", names, [2] * len(names), 5000, 40000, None, 3)


def test_oversized_bgzf_isize_is_rejected(tmp_path):
    """A BGZF trailer is file content: a member that claims more than 64 KiB of data is corrupt,
    not a reason to allocate what it says (up to 4 GiB per member)."""
    from sai_amd.utils.native_vcf import load_dosage, scan_first_last

    rng = np.random.default_rng(11)
    path = tmp_path / "big.vcf.gz"
    names = write_vcf(path, rng, 300, 5, gz="bgzf")
    raw = bytearray(path.read_bytes())
    bsize = (raw[16] | raw[17] << 8) + 1  # first member: BC subfield right after the 12-byte header + 4
    assert raw[12:14] == b"BC"
    raw[bsize - 4 : bsize] = (0xFFFFFFF0).to_bytes(4, "little")  # ISIZE of the first member
    path.write_bytes(bytes(raw))
    with pytest.raises((ValueError, OSError), match="BGZF"):
        scan_first_last(str(path), "21")
    with pytest.raises((ValueError, OSError), match="BGZF"):
        load_dosage(str(path), "21", names, [2] * len(names), None, None, None, 2)


@pytest.mark.parametrize("fixture,chroms", [("tests/data/test.with.outgroup.vcf.gz", ["1", "2"]),
                                            ("tests/data/test.mixed.ploidy.data.vcf.gz", ["20", "21", "22", "X"])])  # fmt: skip
def test_htslib_written_index_equals_unindexed_copy(in_repo_root, tmp_path, fixture, chroms):
    """The reference's fixtures ship with their real (htslib-written, bgzip-compressed) .tbi files:
    scans and region loads through them equal those on a copy of the VCF that has no index."""
    import shutil

    from sai_amd.utils.native_vcf import load_dosage, scan_first_last

    assert (in_repo_root / (fixture + ".tbi")).exists()
    plain = tmp_path / "copy.vcf.gz"
    shutil.copy(fixture, plain)
    with gzip.open(fixture, "rt") as f:
        header = next(line for line in f if line.startswith("#CHROM")).rstrip("\n").split("\t")[9:]
    pick = header[:40]
    for chrom in chroms:
        span = scan_first_last(fixture, chrom)
        assert span == scan_first_last(str(plain), chrom)
        first, last = span if span[0] is not None else (100, 200)
        for reg in ((None, None), (first, last), (first + 1000, max(first + 1000, last - 1000)), (last, last + 10),
                    (last + 1, None), (1, first), (16384, 16385), (16385, 40000)):  # fmt: skip
            a = load_dosage(fixture, chrom, pick, [2] * len(pick), reg[0], reg[1], None, 4)
            b = load_dosage(str(plain), chrom, pick, [2] * len(pick), reg[0], reg[1], None, 4)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], (chrom, reg)


def test_bgzf_through_zlib_when_libdeflate_is_switched_off(tmp_path):
    """BGZF members are inflated by libdeflate when its runtime library is present (bound with
    dlopen); SAI_NO_LIBDEFLATE=1 keeps zlib.  Both give the same rows; a child interpreter is used
    because the choice is made once per process."""
    import os
    import subprocess
    import sys

    from conftest import ROOT
    from sai_amd.utils.native_vcf import load_dosage

    rng = np.random.default_rng(21)
    path = tmp_path / "z.vcf.gz"
    names = write_vcf(path, rng, 400, 6, gz="bgzf")
    want = load_dosage(str(path), "21", names, [2] * 6, None, None, None, 3)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); from sai_amd.utils.native_vcf import load_dosage; "
        "r = load_dosage(%r, '21', %r, [2] * 6, None, None, None, 3); np.save(%r, r[1]); print(len(r[0]))"
        % (str(ROOT), str(path), names, str(tmp_path / "z.npy"))
    )
    res = subprocess.run([sys.executable, "-c", code], env={**os.environ, "SAI_NO_LIBDEFLATE": "1"}, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert int(res.stdout.strip()) == len(want[0]) and np.array_equal(np.load(tmp_path / "z.npy"), want[1])


def test_ancestral_allele_table_semantics(tmp_path):
    """The BED loader (sorted arrays, parsed by several threads): unsorted files, a position listed
    twice (the later line wins), alleles of several characters, runs of blanks and tabs, CRLF, extra
    columns, other chromosomes, plain and gzip -- the native reader against the Python statement of
    the rules; a line with one to three columns is an error."""
    import gzip

    from sai_amd.utils.native_vcf import load_dosage

    rng = np.random.default_rng(12)
    path = tmp_path / "a.vcf"
    names = write_vcf(path, rng, 600, 7)
    reg_pos, reg_ref, reg_alt = [], [], []
    for line in open(path):
        if line.startswith("21\t"):
            f = line.split("\t")
            reg_pos.append(int(f[1])), reg_ref.append(f[3]), reg_alt.append(f[4].split(",")[0])
    lines = []
    for p, r, a in zip(reg_pos, reg_ref, reg_alt):
        u = rng.random()
        allele = r if u < 0.4 else (a if u < 0.7 else ("N" if u < 0.8 else r + a))
        sep = ["\t", " ", "  \t ", "\t\t"][int(rng.integers(4))]
        extra = "" if rng.random() < 0.7 else sep + "extra" + sep + "columns"
        lines.append(f"21{sep}{p - 1}{sep}{p}{sep}{allele}{extra}" + ("\r\n" if rng.random() < 0.2 else "\n"))
        if rng.random() < 0.15:  # listed again with another allele: the later line counts
            lines.append(f"21\t{p - 1}\t{p}\t{a if allele != a else r}\n")
        if rng.random() < 0.1:
            lines.append(f"22\t{p - 1}\t{p}\t{a}\n")
        if rng.random() < 0.05:
            lines.append("\n")
    # padding so that the parse is cut into several pieces, then shuffled: an unsorted file
    lines += [f"7\t{i}\t{i + 1}\tA\n" for i in range(60000)]
    order = rng.permutation(len(lines))
    dup_safe = [lines[i] for i in order]
    for name, body in (("sorted", lines), ("shuffled", dup_safe)):
        bed = tmp_path / f"{name}.bed"
        with open(bed, "w", newline="") as f:
            f.write("".join(body))
        bed_gz = tmp_path / f"{name}.bed.gz"
        with gzip.open(bed_gz, "wt", newline="") as f:
            f.write("".join(body))
        for start, end in ((None, None), (reg_pos[50], reg_pos[400])):
            want = python_reader(path, "21", names[:3], 2, start, end, str(bed))
            for b in (bed, bed_gz):
                got = load_dosage(str(path), "21", names[:3], [2, 2, 2], start, end, str(b), 3)
                assert got[0].tolist() == want[0].tolist(), (name, start)
                assert np.array_equal(got[1], want[1]) and got[2] == want[2] and got[3] == want[3]
    bad = tmp_path / "bad.bed"
    open(bad, "w").write("21\t5\t6\tA\n21\t7\t8\n")
    with pytest.raises(ValueError, match="fewer than 4 columns"):
        load_dosage(str(path), "21", names[:3], [2, 2, 2], None, None, str(bad), 2)
