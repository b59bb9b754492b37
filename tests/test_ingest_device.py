"""The GPU ingest (sai_vcf_stream_* + sai_tokenize_gt, sai_amd.utils.device_vcf): the text of a VCF
region crosses PCIe as it is and is tokenised on the GPU -- same positions, same dosage bytes, same
errors as the host tokenizer (sai_vcf_load), which test_ingest_native.py pins to the Python
statement of the reference's rules."""

import numpy as np
import pytest

from test_ingest_native import write_tbi, write_vcf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from sai_amd.engine import Engine

    return Engine.get(0)


def untile(pop):
    raw = pop.tiles.cpu().numpy()
    n_tiles = (pop.n_sites + 63) // 64
    return raw.reshape(n_tiles, pop.n_ind, 64).transpose(0, 2, 1).reshape(-1, pop.n_ind)[: pop.n_sites]


@pytest.mark.parametrize("gz,crlf", [(False, False), (True, False), (False, True), ("bgzf", False), ("bgzf", True)])
def test_device_reader_equals_host_reader(eng, tmp_path, gz, crlf, monkeypatch):
    from sai_amd.utils.device_vcf import load_dosage_device
    from sai_amd.utils.native_vcf import load_dosage
    from sai_amd.utils.vcf import read_region

    rng = np.random.default_rng(41 + bool(gz) + 2 * crlf)
    path = tmp_path / ("t.vcf.gz" if gz else "t.vcf")
    names = write_vcf(path, rng, 500, 37, gz=gz, crlf=crlf)
    bed = tmp_path / "anc.bed"
    reg = read_region(str(path), "21", names[:1])
    with open(bed, "w") as f:
        for p, r, a in zip(reg.pos, reg.ref, reg.alt):
            u = rng.random()
            if u >= 0.3:
                f.write(f"21\t{p - 1}\t{p}\t{r if u < 0.6 else (a if u < 0.9 else '-')}\n")
    pick = [names[i] for i in rng.permutation(37)[:29]]
    ploidies = [int(rng.choice([1, 2, 2, 3, 4])) for _ in pick]
    for batch_env in ("20000", None):  # tiny reader batches, then the default
        if batch_env:
            monkeypatch.setenv("SAI_VCF_BATCH_BYTES", batch_env)
        else:
            monkeypatch.delenv("SAI_VCF_BATCH_BYTES")
        for start, end in ((None, None), (500, 9000), (9001, 9001), (10**7, None)):
            for anc in (None, str(bed)):
                for cap in (1 << 16, None):
                    pos, dos, nm, na = load_dosage_device(eng, str(path), "21", pick, ploidies, start, end, anc, 3, cap)
                    want = load_dosage(str(path), "21", pick, ploidies, start, end, anc, 2)
                    assert pos.dtype == np.int32 and pos.tolist() == want[0].tolist()
                    assert tuple(dos.shape) == want[1].shape and np.array_equal(dos.cpu().numpy(), want[1])
                    assert nm == want[2]
                    if anc and want[2]:
                        assert na == want[3]


def test_device_reader_edge_lines_and_errors(eng, tmp_path):
    from sai_amd.utils.device_vcf import load_dosage_device
    from sai_amd.utils.native_vcf import load_dosage

    head = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ta\tb\tc\n"
    ok = tmp_path / "edge.vcf"
    ok.write_text(
        head
        + "1\t10\t.\tA\tT\t.\t.\t.\tGT\t0|1\t1/1\t.\n"          # a lone '.'
        + "1\t20\t.\tA\tT\t.\t.\t.\tGT\t\t1\t\n"                # empty fields, the line ends in a tab
        + "1\t30\t.\tA\tT\t.\t.\t.\tDP:GT\t5:1|1\t7\t3:0|0|1\n"  # a field without its GT sub-field, extra alleles
        + "1\t40\t.\tA\tT,G\t.\t.\t.\tGT:DP\t2|10:4\t0|2\t1\n"    # allele indices 2 and 10
        + "1\t50\t.\tA\tT\t.\t.\t.\tGT\t0|1\t1|1\t0/0"            # no newline at the end of the file
    )
    for ploidies in ([2, 2, 2], [1, 3, 4]):
        got = load_dosage_device(eng, str(ok), "1", ["a", "b", "c"], ploidies)
        want = load_dosage(str(ok), "1", ["a", "b", "c"], ploidies)
        assert got[0].tolist() == want[0].tolist() == [10, 20, 30, 40, 50]
        assert np.array_equal(got[1].cpu().numpy(), want[1])
    # a subset in another order, the last VCF column not selected
    got = load_dosage_device(eng, str(ok), "1", ["b", "a"], [2, 2])
    want = load_dosage(str(ok), "1", ["b", "a"], [2, 2])
    assert np.array_equal(got[1].cpu().numpy(), want[1])
    for body, msg in (("1\t10\t.\tA\tT\t.\t.\t.\tGT\t0|1\t0|x\t0|0\n", "unparsable genotype"),
                      ("1\t10\t.\tA\tT\t.\t.\t.\tGT\t0|1\t0|1\n", "too few sample columns"),
                      ("1\t10\t.\tA\tT\t.\t.\t.\tGT\t0|1\t99|99\t0|0\n", "int8")):  # fmt: skip
        bad = tmp_path / "bad.vcf"
        bad.write_text(head + "1\t5\t.\tA\tT\t.\t.\t.\tGT\t0|0\t0|0\t0|0\n" + body)
        with pytest.raises(ValueError, match=msg):
            load_dosage(str(bad), "1", ["a", "b", "c"], [2, 2, 2])
        with pytest.raises(ValueError, match=msg):
            load_dosage_device(eng, str(bad), "1", ["a", "b", "c"], [2, 2, 2])
    # a line longer than the staging buffer: the stream refuses, the caller sees a ValueError
    rng = np.random.default_rng(2)
    wide = tmp_path / "wide.vcf"
    wnames = write_vcf(wide, rng, 3, 30000, chroms=("21",))
    with pytest.raises(ValueError, match="staging buffer"):
        load_dosage_device(eng, str(wide), "21", wnames[:2], [2, 2], buffer_bytes=1 << 16)
    got = load_dosage_device(eng, str(wide), "21", wnames[5:9], [2] * 4)  # the default buffer holds it
    assert np.array_equal(got[1].cpu().numpy(), load_dosage(str(wide), "21", wnames[5:9], [2] * 4)[1])


def test_tabix_region_through_the_device_reader(eng, tmp_path):
    from sai_amd.utils.device_vcf import load_dosage_device
    from sai_amd.utils.native_vcf import load_dosage

    rng = np.random.default_rng(5)
    path = tmp_path / "i.vcf.gz"
    names = write_vcf(path, rng, 1500, 7, gz="bgzf")
    write_tbi(path)
    for chrom in ("7", "22", "nope"):
        for reg in ((1, 10**9), (5000, 40000), (16385, 32768), (70000, 70010), (33000, None)):
            got = load_dosage_device(eng, str(path), chrom, names, [2] * 7, reg[0], reg[1])
            want = load_dosage(str(path), chrom, names, [2] * 7, reg[0], reg[1], None, 3)
            assert got[0].tolist() == want[0].tolist() and np.array_equal(got[1].cpu().numpy(), want[1]) and got[2] == want[2]


def test_region_of_an_indexed_bgzip_file_is_a_seek_on_the_gpu_route(eng, tmp_path, monkeypatch):
    """A region of a bgzip file with a usable .tbi takes the GPU-inflate route too, and only the members
    that hold the region cross PCIe: first / middle / last / empty regions, regions that start in the
    middle of a member, one-record regions -- positions, dosage bytes and match counts of the
    host-inflating stream (which seeks through the same index) and of the host reader; both index
    modes; a stale index means the full pass, same answer."""
    import os

    from sai_amd.utils import device_vcf
    from sai_amd.utils.native_vcf import load_dosage

    rng = np.random.default_rng(31)
    path = tmp_path / "r.vcf.gz"
    names = write_vcf(path, rng, 6000, 150, gz="bgzf")  # ~4 MB of text: some 70 members over three chromosomes
    write_tbi(path)
    pick, ploidies = names[10:90], [2] * 80
    whole = {c: load_dosage(str(path), c, pick, ploidies, None, None, None, 2)[0] for c in ("7", "21", "22")}
    file_bytes = os.path.getsize(path)
    seen_partial = 0
    for chrom, pos in whole.items():
        n = len(pos)
        assert n > 1500
        regions = [(1, int(pos[40])), (int(pos[n // 2]), int(pos[n // 2 + 300])), (int(pos[-200]), int(pos[-1]) + 5000),
                   (int(pos[-1]) + 1, int(pos[-1]) + 10), (int(pos[n // 3]) + 1, int(pos[n // 3 + 1]) - 1),
                   (int(pos[700]), int(pos[700])), (int(pos[n // 4]), None), (1, 10**9)]  # fmt: skip
        for start, end in regions:
            want = load_dosage(str(path), chrom, pick, ploidies, start, end, None, 2)
            for mode in ("heads", "text"):
                monkeypatch.setenv("SAI_AMD_BGZF_INDEX", mode)
                for cap in (1 << 17, None):
                    got = device_vcf.load_dosage_device(eng, str(path), chrom, pick, ploidies, start, end, None, 4, cap)
                    last = eng._inflate_state["last"]
                    assert got[0].tolist() == want[0].tolist() and got[2] == want[2], (chrom, start, end, mode, cap)
                    assert np.array_equal(got[1].cpu().numpy(), want[1])
                    if end is not None and end < pos[-1] and start > pos[0]:  # an inner region: a small part of the file
                        assert 0 < last["comp_bytes"] < file_bytes // 3 and last["file_stop"] >= last["file_begin"] > 0
                        seen_partial += 1
            monkeypatch.delenv("SAI_AMD_BGZF_INDEX")
            monkeypatch.setenv("SAI_AMD_GPU_INFLATE", "0")  # the host-inflating stream, seeking through the same index
            host = device_vcf.load_dosage_device(eng, str(path), chrom, pick, ploidies, start, end, None, 4, 1 << 18)
            monkeypatch.delenv("SAI_AMD_GPU_INFLATE")
            assert host[0].tolist() == want[0].tolist() and np.array_equal(host[1].cpu().numpy(), want[1])
    assert seen_partial >= 12
    # a chromosome the index does not know: nothing is read, nothing is found
    got = device_vcf.load_dosage_device(eng, str(path), "nope", pick, ploidies, 1, 10**6)
    assert got[0].size == 0 and eng._inflate_state["last"]["members"] == 0
    # a stale index (older than its file) is ignored: the full pass, the same records
    old = os.stat(path).st_mtime - 100
    os.utime(str(path) + ".tbi", (old, old))
    pos = whole["21"]
    got = device_vcf.load_dosage_device(eng, str(path), "21", pick, ploidies, int(pos[900]), int(pos[950]))
    want = load_dosage(str(path), "21", pick, ploidies, int(pos[900]), int(pos[950]), None, 2)
    assert got[0].tolist() == want[0].tolist() and np.array_equal(got[1].cpu().numpy(), want[1])
    assert eng._inflate_state["last"]["file_begin"] == 0 and eng._inflate_state["last"]["comp_bytes"] > file_bytes // 2


@pytest.mark.parametrize("vcf,chrom,cfgfile,anc", [
    ("tests/data/test.with.outgroup.vcf.gz", "1", "tests/data/test.with.outgroup.config.yaml", "tests/data/test.with.outgroup.anc.alleles"),
    ("tests/data/test.mixed.ploidy.data.vcf.gz", "21", "tests/data/test_mixed_ploidy.config.yaml", "tests/data/test.mixed.ploidy.data.anc.alleles"),
    ("tests/data/example.vcf", "21", "tests/data/example.u_and_q.config.yaml", None),
])  # fmt: skip
def test_read_data_device_and_score_equal_the_host_path(eng, in_repo_root, tmp_path, monkeypatch, vcf, chrom, cfgfile, anc):
    """The reference's fixtures: every population block read on the GPU equals the host reader's
    matrix, and `score` writes byte-identical files with either reader."""
    import os

    from sai_amd.sai import load_config, score
    from sai_amd.utils.read_data import read_data_device
    from sai_amd.utils.read_data import read_dosage_data as read_data

    if not os.path.exists(cfgfile):
        pytest.skip("fixture config not present")
    cfg = load_config(cfgfile)
    kw = dict(vcf_file=vcf, chr_name=chrom, ploidy_config=cfg.ploidies, ref_ind_file=cfg.populations.get_population("ref"),
              tgt_ind_file=cfg.populations.get_population("tgt"), src_ind_file=cfg.populations.get_population("src"),
              out_ind_file=cfg.populations.get_population("outgroup"), anc_allele_file=anc)  # fmt: skip
    host = read_data(**kw)
    dev, pos_dev = read_data_device(eng, **kw)
    for group in ("ref", "tgt", "src", "outgroup"):
        hd, dd = host[group][0], dev[group][0]
        assert (hd is None) == (dd is None)
        for pop in hd or {}:
            assert hd[pop].POS.tolist() == dd[pop].POS.tolist() == pos_dev.cpu().numpy().tolist()
            assert np.array_equal(untile(dd[pop].GT), hd[pop].GT)
    outs = {}
    for mode in ("device", "host"):
        monkeypatch.setenv("SAI_AMD_INGEST", mode)
        out = tmp_path / f"{mode}.tsv"
        score(vcf_file=vcf, chr_name=chrom, win_len=20000, win_step=10000, anc_allele_file=anc, output_file=str(out),
              config=cfgfile, num_workers=1)  # fmt: skip
        outs[mode] = {p.suffixes[-2] if len(p.suffixes) > 1 else "": p.read_text() for p in tmp_path.glob(f"{mode}*")}
    assert outs["device"] == outs["host"] and len(outs["device"][""].splitlines()) > 1


def test_a_sample_in_populations_of_different_ploidy(eng, tmp_path):
    """The reference reads every population on its own, with its own ploidy (utils.py:123-138), so one
    sample may sit in a diploid reference population and a tetraploid target population.  The streaming
    reader maps a VCF column to one slot per pass and tokenises such a sample in a second pass: same
    blocks as the host reader (which reads per population), also when a population's samples come from
    different passes; and `score` runs on it with either reader."""
    from sai_amd.configs import PloidyConfig
    from sai_amd.utils.read_data import read_data_device
    from sai_amd.utils.read_data import read_dosage_data as read_data

    rng = np.random.default_rng(77)
    vcf = tmp_path / "m.vcf.gz"
    names = write_vcf(vcf, rng, 400, 12, gz="bgzf", chroms=("21",))
    files = {}
    for group, pops in (("ref", {"A": names[0:5], "B": names[3:7]}), ("tgt", {"T": names[4:10]}), ("src", {"S": names[10:12]})):
        files[group] = tmp_path / f"{group}.txt"
        files[group].write_text("".join(f"{pop}\t{n}\n" for pop, members in pops.items() for n in members))
    # names[3] and names[4] are read at ploidy 2 (A), 4 (B) and, names[4], 3 (T): three passes for names[4]
    pc = PloidyConfig({"ref": {"A": 2, "B": 4}, "tgt": {"T": 3}, "src": {"S": 2}})
    kw = dict(vcf_file=str(vcf), chr_name="21", ploidy_config=pc, ref_ind_file=str(files["ref"]), tgt_ind_file=str(files["tgt"]),
              src_ind_file=str(files["src"]))  # fmt: skip
    host = read_data(**kw)
    dev, pos_dev = read_data_device(eng, **kw)
    n_blocks = 0
    for group in ("ref", "tgt", "src"):
        for pop, cd in host[group][0].items():
            got = dev[group][0][pop]
            assert cd.POS.tolist() == got.POS.tolist() == pos_dev.cpu().numpy().tolist()
            assert np.array_equal(untile(got.GT), cd.GT), (group, pop)
            n_blocks += 1
    assert n_blocks == 4
    a, b = host["ref"][0]["A"].GT, host["ref"][0]["B"].GT
    assert not np.array_equal(a[:, 3], b[:, 0])  # the shared sample really reads differently at the two ploidies


import os  # noqa: E402


@pytest.mark.parametrize("seed", range(int(os.environ.get("SAI_INGEST_FUZZ", "6"))))
def test_device_reader_fuzz(eng, tmp_path, seed, monkeypatch):
    """Random files (size, samples, container, line ends), random selections, ploidies, regions,
    staging-buffer and batch sizes: the GPU reader against the host reader.  SAI_INGEST_FUZZ=300 was
    run once on the GPU box."""
    from sai_amd.utils.device_vcf import load_dosage_device
    from sai_amd.utils.native_vcf import load_dosage

    rng = np.random.default_rng(1000 + seed)
    gz = [False, True, "bgzf"][int(rng.integers(3))]
    n_samples = int(rng.integers(1, 90))
    path = tmp_path / ("f.vcf.gz" if gz else "f.vcf")
    names = write_vcf(path, rng, int(rng.integers(1, 400)), n_samples, gz=gz, crlf=bool(rng.integers(2)))
    if rng.random() < 0.5:
        monkeypatch.setenv("SAI_VCF_BATCH_BYTES", str(int(rng.integers(3000, 200000))))
    k = int(rng.integers(1, n_samples + 1))
    pick = [names[i] for i in rng.permutation(n_samples)[:k]]
    ploidies = [int(rng.integers(1, 5)) for _ in pick]
    chrom = str(rng.choice(["7", "21", "22"]))
    for _ in range(3):
        start = None if rng.random() < 0.4 else int(rng.integers(1, 20000))
        end = None if start is None or rng.random() < 0.3 else start + int(rng.integers(0, 20000))
        cap = None if rng.random() < 0.5 else int(rng.integers(1 << 16, 1 << 18))
        got = load_dosage_device(eng, str(path), chrom, pick, ploidies, start, end, None, int(rng.integers(1, 7)), cap)
        want = load_dosage(str(path), chrom, pick, ploidies, start, end, None, 2)
        assert got[0].tolist() == want[0].tolist() and got[2] == want[2]
        assert np.array_equal(got[1].cpu().numpy(), want[1])


def test_bgzip_is_inflated_on_the_gpu(eng, tmp_path, monkeypatch):
    """A bgzip file read without a region seek takes the sai_inflate_bgzf route (compressed bytes
    over PCIe, text indexed once on the host, tokenised in HBM): many small batches with a carried
    line between them, a last line without a newline, the same answer as the host-inflating stream
    and as the host reader; a damaged member is reported, not followed."""
    import zlib

    from sai_amd.utils import device_vcf
    from sai_amd.utils.native_vcf import load_dosage

    rng = np.random.default_rng(9)
    path = tmp_path / "big.vcf.gz"
    names = write_vcf(path, rng, 2500, 61, gz="bgzf")
    calls = {"n": 0}
    real = device_vcf._load_bgzf_device

    def spy(*a, **k):
        got = real(*a, **k)
        calls["n"] += got is not None
        return got

    monkeypatch.setattr(device_vcf, "_load_bgzf_device", spy)
    pick = [names[i] for i in rng.permutation(61)[:40]]
    ploidies = [int(rng.choice([1, 2, 2, 4])) for _ in pick]
    for chrom in ("7", "21", "22"):
        for start, end in ((None, None), (300, 20000)):
            want = load_dosage(str(path), chrom, pick, ploidies, start, end, None, 2)
            for cap in (1 << 16, 1 << 18, None):
                before = calls["n"]
                got = device_vcf.load_dosage_device(eng, str(path), chrom, pick, ploidies, start, end, None, 4, cap)
                assert calls["n"] == before + 1
                assert got[0].tolist() == want[0].tolist() and got[2] == want[2]
                assert np.array_equal(got[1].cpu().numpy(), want[1])
            monkeypatch.setenv("SAI_AMD_GPU_INFLATE", "0")
            before = calls["n"]
            host = device_vcf.load_dosage_device(eng, str(path), chrom, pick, ploidies, start, end, None, 4, 1 << 18)
            monkeypatch.delenv("SAI_AMD_GPU_INFLATE")
            assert calls["n"] == before and host[0].tolist() == want[0].tolist() and np.array_equal(host[1].cpu().numpy(), want[1])
    # the last record line without its newline (re-pack the text in members of odd sizes)
    import gzip
    import struct

    text = gzip.open(path, "rb").read().rstrip(b"\n")

    def member(chunk):
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = comp.compress(chunk) + comp.flush()
        head = b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(raw) + 8 - 1)
        return head + raw + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))

    odd = tmp_path / "odd.vcf.gz"
    with open(odd, "wb") as f:
        o = 0
        while o < len(text):
            step = int(rng.integers(1, 65281))
            f.write(member(text[o : o + step]))
            o += step
        f.write(member(b""))
    want = load_dosage(str(odd), "22", pick, ploidies, None, None, None, 2)
    for cap in (1 << 16, None):
        before = calls["n"]
        got = device_vcf.load_dosage_device(eng, str(odd), "22", pick, ploidies, None, None, None, 3, cap)
        assert calls["n"] == before + 1 and got[0].tolist() == want[0].tolist() and np.array_equal(got[1].cpu().numpy(), want[1])
    # both ways of indexing (line heads extracted on the GPU / the whole text copied back) agree
    for mode in ("heads", "text"):
        monkeypatch.setenv("SAI_AMD_BGZF_INDEX", mode)
        for cap in (1 << 16, None):
            got = device_vcf.load_dosage_device(eng, str(odd), "22", pick, ploidies, None, None, None, 3, cap)
            assert got[0].tolist() == want[0].tolist() and np.array_equal(got[1].cpu().numpy(), want[1])
    monkeypatch.delenv("SAI_AMD_BGZF_INDEX")
    # a record whose INFO column pushes FORMAT beyond the reach of the line heads: the file is indexed from the text
    lines = text.split(b"\n")
    k = next(i for i, ln in enumerate(lines) if ln.startswith(b"22\t"))
    f = lines[k].split(b"\t")
    f[7] = b"NOTE=" + b"x" * 6000
    lines[k] = b"\t".join(f)
    far = tmp_path / "far.vcf.gz"
    with open(far, "wb") as fh:
        blob = b"\n".join(lines) + b"\n"
        for o in range(0, len(blob), 60000):
            fh.write(member(blob[o : o + 60000]))
        fh.write(member(b""))
    seen = []
    real_inner = real

    def spy_modes(*a, **k):
        seen.append(k.get("index_from"))
        return real_inner(*a, **k)

    monkeypatch.setattr(device_vcf, "_load_bgzf_device", spy_modes)
    want_far = load_dosage(str(far), "22", pick, ploidies, None, None, None, 2)
    got = device_vcf.load_dosage_device(eng, str(far), "22", pick, ploidies)
    assert seen == [None, "text"] and got[0].tolist() == want_far[0].tolist() and np.array_equal(got[1].cpu().numpy(), want_far[1])
    monkeypatch.setattr(device_vcf, "_load_bgzf_device", spy)
    # damage in the middle of the file: the kernel's status or the host's CRC check stops the read
    raw = bytearray(open(path, "rb").read())
    raw[len(raw) // 2] ^= 0x55
    bad = tmp_path / "bad.vcf.gz"
    open(bad, "wb").write(raw)
    with pytest.raises(ValueError, match="BGZF|corrupt"):
        device_vcf.load_dosage_device(eng, str(bad), "22", pick, ploidies)
    # and the buffers are in order for the next call
    got = device_vcf.load_dosage_device(eng, str(path), "22", pick, ploidies)
    want = load_dosage(str(path), "22", pick, ploidies, None, None, None, 2)
    assert got[0].tolist() == want[0].tolist() and np.array_equal(got[1].cpu().numpy(), want[1])


def test_chromosome_scan_of_a_bgzip_file_runs_on_the_gpu(eng, tmp_path, monkeypatch):
    """ChunkGenerator's first / last position of a chromosome: the GPU-inflate pass without a
    tokenizer gives what the host scan gives (file order, first contiguous run, absent chromosome),
    plain and gzip files stay with the host."""
    from sai_amd.utils import device_vcf, native_vcf

    rng = np.random.default_rng(21)
    path = tmp_path / "s.vcf.gz"
    write_vcf(path, rng, 1500, 30, gz="bgzf")
    calls = {"n": 0}
    real = device_vcf.scan_first_last_device

    def spy(*a, **k):
        got = real(*a, **k)
        calls["n"] += got is not None
        return got

    monkeypatch.setattr(device_vcf, "scan_first_last_device", spy)
    for chrom in ("7", "21", "22", "X"):
        got = native_vcf.scan_first_last(str(path), chrom)
        monkeypatch.setenv("SAI_AMD_INGEST", "host")
        want = native_vcf.scan_first_last(str(path), chrom)
        monkeypatch.delenv("SAI_AMD_INGEST")
        assert got == want, chrom
    assert calls["n"] == 4
    plain = tmp_path / "p.vcf"
    write_vcf(plain, rng, 200, 5)
    gz = tmp_path / "g.vcf.gz"
    write_vcf(gz, rng, 200, 5, gz=True)
    before = calls["n"]
    assert native_vcf.scan_first_last(str(plain), "21")[0] is not None and native_vcf.scan_first_last(str(gz), "21")[0] is not None
    assert calls["n"] == before
    # with a usable index the host reads two records and the GPU pass over the whole file is not started;
    # an index older than its file is not trusted and the GPU scans again
    spans = {chrom: native_vcf.scan_first_last(str(path), chrom) for chrom in ("7", "21", "22", "X")}
    before = calls["n"]
    write_tbi(path)
    assert {chrom: native_vcf.scan_first_last(str(path), chrom) for chrom in spans} == spans and calls["n"] == before
    st = os.stat(path)
    os.utime(str(path) + ".tbi", (st.st_atime - 100, st.st_mtime - 100))
    assert native_vcf.scan_first_last(str(path), "21") == spans["21"] and calls["n"] == before + 1
