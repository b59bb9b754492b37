"""sai_inflate_bgzf: BGZF members inflated on the GPU, one wavefront per member, against zlib.

Every DEFLATE block type (stored, fixed, dynamic), several blocks per member, matches at the far end
of the window, run-length matches, codes longer than the primary look-up tables, members of 0 bytes
and of exactly 64 KiB, unaligned placements -- and corrupt members, which must end with a non-zero
status (bad DEFLATE, wrong size, or text that fails the trailer's CRC-32) and leave everything outside
their own output range untouched."""

import ctypes as C
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MEMBER = np.dtype([("data_off", "<i8"), ("out_off", "<i8"), ("data_len", "<u4"), ("isize", "<u4"), ("crc", "<u4"), ("reserved", "<u4")])


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from sai_amd.engine import Engine

    return Engine.get(0)


def deflate(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = b""
    for i in range(0, len(data), flush_every):  # a sync flush ends the block and adds an empty stored one
        out += c.compress(data[i : i + flush_every]) + c.flush(zlib.Z_SYNC_FLUSH if (i // flush_every) % 2 else zlib.Z_FULL_FLUSH)
    return out + c.flush()


def run(eng, streams, texts, rng=None, gap=0, text_gap=0):
    """Inflate `streams` (raw deflate) placed one after the other (+ `gap` random bytes between them)
    into a text buffer with `text_gap` guard bytes between the outputs.  Returns (status, outputs,
    guards_intact)."""
    import torch

    comp, table, off, out_off = bytearray(), np.zeros(len(streams), dtype=MEMBER), 0, text_gap
    for i, (s, t) in enumerate(zip(streams, texts)):
        pad = bytes(rng.integers(0, 256, size=gap, dtype=np.uint8)) if (rng is not None and gap) else b""
        comp += pad
        table[i] = (len(comp), out_off, len(s), len(t), zlib.crc32(t), 0)
        comp += s
        out_off += len(t) + text_gap
    comp += b"\0" * (-len(comp) % 4 + 4)
    n_text = out_off + 8
    d_comp = torch.from_numpy(np.frombuffer(bytes(comp), dtype=np.uint8).copy()).to(eng.device)
    d_tab = torch.from_numpy(table.view(np.uint8).copy()).to(eng.device)
    d_text = torch.full((n_text,), 0xA5, dtype=torch.uint8, device=eng.device)
    d_stat = torch.full((len(streams),), -1, dtype=torch.int32, device=eng.device)
    from sai_amd import _ffi

    _ffi.check(eng.lib.sai_inflate_bgzf(eng.ctx, C.c_void_p(d_comp.data_ptr()), d_comp.numel(), C.c_void_p(d_tab.data_ptr()),
                                        len(streams), C.c_void_p(d_text.data_ptr()), n_text, C.c_void_p(d_stat.data_ptr()), None))  # fmt: skip
    torch.cuda.synchronize()
    text = d_text.cpu().numpy()
    outs, guards = [], True
    prev_end = 0
    for row in table:
        o, n = int(row["out_off"]), int(row["isize"])
        guards = guards and bool((text[prev_end:o] == 0xA5).all())
        outs.append(text[o : o + n].tobytes())
        prev_end = o + n
    guards = guards and bool((text[prev_end:] == 0xA5).all())
    return d_stat.cpu().numpy(), outs, guards


def vcf_like(rng, n):
    calls = np.array([b"0|0", b"0|1", b"1|0", b"1|1", b".|."])
    out, pos = bytearray(), 0
    while len(out) < n:
        pos += int(rng.integers(1, 900))
        row = calls[rng.choice(5, size=400, p=[0.8, 0.07, 0.07, 0.055, 0.005])]
        out += b"21\t%d\trs%d\tA\tG\t.\tPASS\tAC=%d;AN=800\tGT\t" % (pos, pos, int(rng.integers(0, 800))) + b"\t".join(row) + b"\n"
    return bytes(out[:n])


def test_every_block_type_and_shape(eng):
    rng = np.random.default_rng(3)
    texts, streams = [], []

    def add(t, **kw):
        texts.append(t)
        streams.append(deflate(t, **kw))

    text = vcf_like(rng, 65536)
    add(text)  # dynamic blocks, a full 64 KiB member
    add(text[:65280], level=1)
    add(text[:1], level=9)
    add(b"")  # bgzip's EOF member: one empty fixed block
    add(text[:300], strategy=zlib.Z_FIXED)
    add(text[:40000], strategy=zlib.Z_FIXED)
    add(text[:50000], strategy=zlib.Z_HUFFMAN_ONLY)  # literals only, no distance codes
    add(text[:50000], strategy=zlib.Z_RLE)  # distance 1 only
    add(bytes(rng.integers(0, 256, size=65536, dtype=np.uint8)))  # incompressible: stored blocks
    add(bytes(rng.integers(0, 256, size=70, dtype=np.uint8)), level=0)
    add(bytes(rng.integers(0, 256, size=65535, dtype=np.uint8)), level=0)  # one stored block longer than the window
    add(b"\0" * 65536)  # run-length matches of 258
    add(b"ab" * 30000)
    add(bytes(rng.integers(0, 4, size=60000, dtype=np.uint8)))  # short codes
    add(bytes(rng.integers(0, 256, size=2000, dtype=np.uint8)) * 30)  # long matches at distance 2000
    period = bytes(rng.integers(0, 256, size=32768, dtype=np.uint8))
    add(period + period)  # matches at distance 32768, the far end of the window
    add(period[:20000] * 3 + period[:5536])  # distances beyond the 16 KiB kept in LDS: served from the text in HBM
    mix = bytearray(vcf_like(rng, 65536))
    for _ in range(40):  # far and near matches interleaved
        o = int(rng.integers(0, 30000))
        n = int(rng.integers(3, 300))
        t = int(rng.integers(o + 17000, 65536 - n))
        mix[t : t + n] = mix[o : o + n]
    add(bytes(mix), level=9)
    add(text[:60000], flush_every=7000)  # several blocks per member, empty stored blocks between them
    add(text[:60000], level=9, flush_every=100)
    # skewed symbol frequencies: Huffman codes longer than the 10-bit / 8-bit primary tables
    p = 0.5 ** np.arange(1, 257)
    add(bytes(rng.choice(256, size=65000, p=p / p.sum()).astype(np.uint8)), strategy=zlib.Z_HUFFMAN_ONLY)
    add(bytes(rng.choice(256, size=65000, p=p / p.sum()).astype(np.uint8)))
    for gap, text_gap in ((0, 0), (1, 1), (3, 5), (7, 64)):
        status, outs, guards = run(eng, streams, texts, rng, gap, text_gap)
        assert status.tolist() == [0] * len(texts), (gap, status.tolist())
        for i, (got, want) in enumerate(zip(outs, texts)):
            assert got == want, (gap, i, len(want))
        assert guards


def test_many_members_random(eng):
    rng = np.random.default_rng(11)
    base = vcf_like(rng, 1 << 20)
    texts, streams = [], []
    for i in range(600):
        n = int(rng.integers(0, 65537)) if i % 7 else 65280
        o = int(rng.integers(0, len(base) - n + 1))
        t = base[o : o + n]
        if i % 11 == 0:
            t = bytes(rng.integers(0, 1 + int(rng.integers(1, 256)), size=n, dtype=np.uint8))
        texts.append(t)
        streams.append(deflate(t, level=int(rng.integers(0, 10)),
                               strategy=[zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_FIXED, zlib.Z_RLE][int(rng.integers(4))]))  # fmt: skip
    status, outs, guards = run(eng, streams, texts, rng, 2, 3)
    assert not status.any() and guards
    assert all(a == b for a, b in zip(outs, texts))


def test_corrupt_members_are_flagged_not_followed(eng):
    """Bit flips, truncation, a wrong ISIZE: status != 0 (or, where the damage still is valid DEFLATE
    of the right size, text that fails the CRC -- the host's check), nothing written outside the
    member's own range, and the undamaged neighbours come out right."""
    rng = np.random.default_rng(5)
    text = vcf_like(rng, 65536)
    good = deflate(text)
    streams, texts, expect_bad = [], [], []
    for k in range(120):
        s = bytearray(good)
        kind = k % 4
        if kind == 0:  # flip bits
            for _ in range(int(rng.integers(1, 4))):
                s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(8))
            t = text
        elif kind == 1:  # truncated stream
            s = s[: int(rng.integers(1, len(s)))]
            t = text
        elif kind == 2:  # ISIZE too small / too large
            t = text[: int(rng.integers(0, 65536))] if k % 8 == 2 else text + b"x" * 0
            if len(t) == len(text):
                t = text[:-1]
        else:  # random bytes posing as a stream
            s = bytearray(rng.integers(0, 256, size=int(rng.integers(1, 3000)), dtype=np.uint8))
            t = text[: int(rng.integers(1, 65536))]
        streams += [bytes(s), good]
        texts += [t, text]
        expect_bad.append(kind)
    status, outs, guards = run(eng, streams, texts, rng, 1, 16)
    assert guards
    for i, kind in enumerate(expect_bad):
        bad_status, got, want = int(status[2 * i]), outs[2 * i], texts[2 * i]
        assert int(status[2 * i + 1]) == 0 and outs[2 * i + 1] == text
        if kind in (1, 2):
            assert bad_status != 0, (i, kind)
        else:  # damage that still is DEFLATE of the right size fails the CRC check (status 9)
            assert bad_status != 0 or got == want, (i, kind)
            if bad_status == 0 or got == want:
                assert zlib.crc32(got) == zlib.crc32(want)


def test_argument_checks(eng):
    import torch

    from sai_amd import _ffi

    buf = torch.zeros(64, dtype=torch.uint8, device=eng.device)
    tab = torch.zeros(32, dtype=torch.uint8, device=eng.device)
    st = torch.zeros(1, dtype=torch.int32, device=eng.device)
    p = lambda t, o=0: C.c_void_p(t.data_ptr() + o)  # noqa: E731
    assert eng.lib.sai_inflate_bgzf(eng.ctx, p(buf), 64, p(tab), 0, p(buf), 64, p(st), None) == 0
    assert eng.lib.sai_inflate_bgzf(eng.ctx, None, 64, p(tab), 1, p(buf), 64, p(st), None) == -1
    assert eng.lib.sai_inflate_bgzf(eng.ctx, p(buf, 1), 60, p(tab), 1, p(buf), 64, p(st), None) == -1
    assert eng.lib.sai_inflate_bgzf(eng.ctx, p(buf), 63, p(tab), 1, p(buf), 64, p(st), None) == -1
    # a member whose ranges leave the buffers is refused by the kernel itself
    row = np.zeros(1, dtype=MEMBER)
    row[0] = (0, 0, 4000, 10, 0, 0)
    tab.copy_(torch.from_numpy(row.view(np.uint8).copy()))
    assert eng.lib.sai_inflate_bgzf(eng.ctx, p(buf), 64, p(tab), 1, p(buf), 64, p(st), None) == 0
    torch.cuda.synchronize()
    assert int(st.cpu()[0]) != 0


def test_line_table_kernels(eng):
    """sai_text_line_starts / sai_text_line_heads against their numpy statement: every base
    alignment, CRLF lines, '#' lines, lines with fewer than ten columns, a ninth tab out of reach,
    a batch that ends inside a line, an empty batch, a line capacity that is too small."""
    import torch

    from sai_amd import _ffi

    rng = np.random.default_rng(2)
    body = bytearray(b"##meta\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\ts2\n")
    for i in range(700):
        info = b"X" * int(rng.integers(1, 5000 if i % 97 == 0 else 60))
        line = b"21\t%d\t.\tA\tC\t.\tPASS\t%s\tGT\t" % (i + 1, info) + b"\t".join([b"0|1"] * int(rng.integers(1, 400)))
        if i % 50 == 3:
            line = b"21\t5\tshort"
        if i % 60 == 7:
            line = b""
        body += line + (b"\r\n" if i % 3 == 0 else b"\n")
    body += b"21\t999\tpartial line without its end"
    body = bytes(body)
    for mis in (0, 1, 7, 15, 16, 33):
        for n in (len(body), 0, 1, 4095, 4096, 4097, 20000):
            text = body[:n]
            raw = np.frombuffer(text, dtype=np.uint8)
            ends = np.flatnonzero(raw == 10)
            starts = np.concatenate([[0], ends + 1]).astype(np.int64)
            n_l = len(ends)
            want_info = np.zeros(n_l, dtype=np.int64)
            for i in range(n_l):
                line = text[starts[i] : starts[i + 1] - 1]
                cr = line.endswith(b"\r")
                line = line[:-1] if cr else line
                fixed = 1
                if line and not line.startswith(b"#"):
                    tabs = [k for k, ch in enumerate(line[:4096]) if ch == 9]
                    fixed = tabs[8] + 1 if len(tabs) >= 9 else (len(line) + 1 if len(line) <= 4096 else 4097)
                want_info[i] = fixed | ((1 << 31) if cr else 0)
            cap = n_l + 5
            d_text = torch.zeros((mis + n + 48,), dtype=torch.uint8, device=eng.device)
            d_text[:mis] = 10  # newlines in front of the text must not count
            d_text[mis + n :] = 10  # nor behind it
            if n:
                d_text[mis : mis + n] = torch.from_numpy(raw.copy()).to(eng.device)
            d_starts = torch.full((cap + 1,), -7, dtype=torch.int64, device=eng.device)
            d_info = torch.full((cap,), -7, dtype=torch.int32, device=eng.device)
            d_scr = torch.zeros(((n + 15) // 4096 + 2,), dtype=torch.int32, device=eng.device)
            d_i4 = torch.zeros((4,), dtype=torch.int32, device=eng.device)
            p = lambda t, o=0: C.c_void_p(t.data_ptr() + o)  # noqa: E731
            _ffi.check(eng.lib.sai_text_line_starts(eng.ctx, p(d_text, mis), n, cap, p(d_starts), p(d_info), p(d_scr), p(d_i4), None))
            i4 = d_i4.cpu().tolist()
            assert i4[0] == n_l and i4[2] == 0, (mis, n, i4)
            assert d_starts[: n_l + 1].cpu().tolist() == starts.tolist()
            assert (d_starts[n_l + 1 :] == -7).all()
            got_info = d_info[:n_l].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            assert got_info.tolist() == want_info.tolist(), (mis, n)
            assert i4[1] == (int((want_info & 0x7FFFFFFF).max()) if n_l else 0)
            for hb in (16, 64):
                d_heads = torch.full((max(n_l, 1) * hb + 4,), 0x55, dtype=torch.uint8, device=eng.device)
                _ffi.check(eng.lib.sai_text_line_heads(eng.ctx, p(d_text, mis), n, p(d_starts), n_l, hb, p(d_heads), None))
                heads = d_heads.cpu().numpy()
                assert (heads[n_l * hb :] == 0x55).all()
                for i in range(n_l):
                    chunk = raw[starts[i] : min(starts[i] + hb, starts[i + 1])]
                    want = np.full(hb, 10, dtype=np.uint8)
                    want[: len(chunk)] = chunk
                    assert heads[i * hb : (i + 1) * hb].tolist() == want.tolist(), (mis, n, i)
    # a capacity that is too small is reported and respected
    d_starts = torch.full((4,), -7, dtype=torch.int64, device=eng.device)
    d_info = torch.full((3,), -7, dtype=torch.int32, device=eng.device)
    _ffi.check(eng.lib.sai_text_line_starts(eng.ctx, p(d_text, mis), n, 3, p(d_starts), p(d_info), p(d_scr), p(d_i4), None))
    assert d_i4.cpu().tolist()[2] == 1 and d_starts.cpu().tolist() == starts[:4].tolist()
