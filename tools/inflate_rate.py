"""Rate of sai_inflate_bgzf on VCF-like text (GPU box): python tools/inflate_rate.py [MB of text]"""
import ctypes as C
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import torch  # noqa: E402

from sai_amd import _ffi  # noqa: E402
from sai_amd.engine import Engine  # noqa: E402
from test_inflate_device import MEMBER, deflate  # noqa: E402


def main() -> None:
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 2002
    rng = np.random.default_rng(1)
    calls = np.array([b"0|0", b"0|1", b"1|0", b"1|1", b".|."])
    lines, pos = [], 0
    linked = len(sys.argv) > 3 and sys.argv[3] == "linked"  # every line a copy of the one before with 2 % of the calls redrawn
    row = calls[rng.choice(5, size=n_samples, p=[0.7, 0.1, 0.1, 0.095, 0.005])]
    size = 0
    while size < (8 << 20):
        pos += int(rng.integers(1, 50))
        if linked:
            row = row.copy()
            hit = rng.random(n_samples) < 0.02
            row[hit] = calls[rng.choice(5, size=int(hit.sum()), p=[0.7, 0.1, 0.1, 0.095, 0.005])]
        else:
            row = calls[rng.choice(5, size=n_samples, p=[0.7, 0.1, 0.1, 0.095, 0.005])]
        size += 4 * n_samples + 30
        lines.append(b"1\t%d\t.\tA\tT\t100\tPASS\t.\tGT\t" % pos + b"\t".join(row) + b"\n")
    unit = b"".join(lines)
    text = (unit * (mb * (1 << 20) // len(unit) + 1))[: mb << 20]
    chunks = [text[i : i + 65280] for i in range(0, len(text), 65280)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(16) as ex:
        streams = list(ex.map(deflate, chunks))
    print(f"{n_samples} samples per line; {len(text) / 1e6:.0f} MB of text, {len(chunks)} members, {sum(map(len, streams)) / 1e6:.1f} MB compressed "
          f"(deflated in {time.perf_counter() - t0:.1f} s)", flush=True)  # fmt: skip
    eng = Engine.get(0)
    table = np.zeros(len(chunks), dtype=MEMBER)
    comp, off, out = bytearray(), 0, 0
    for i, (s, t) in enumerate(zip(streams, chunks)):
        table[i] = (len(comp), out, len(s), len(t), zlib.crc32(t), 0)
        comp += s
        out += len(t)
    comp += b"\0" * (-len(comp) % 4 + 4)
    h_comp = torch.from_numpy(np.frombuffer(bytes(comp), dtype=np.uint8).copy()).pin_memory()
    d_tab = torch.from_numpy(table.view(np.uint8).copy()).to(eng.device)
    d_text = torch.empty((out + 8,), dtype=torch.uint8, device=eng.device)
    d_stat = torch.empty((len(chunks),), dtype=torch.int32, device=eng.device)
    d_comp = torch.empty_like(h_comp, device=eng.device)
    for rep in range(4):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        d_comp.copy_(h_comp, non_blocking=True)
        e[1].record()
        _ffi.check(eng.lib.sai_inflate_bgzf(eng.ctx, C.c_void_p(d_comp.data_ptr()), d_comp.numel(), C.c_void_p(d_tab.data_ptr()),
                                            len(chunks), C.c_void_p(d_text.data_ptr()), out + 8, C.c_void_p(d_stat.data_ptr()), None))  # fmt: skip
        e[2].record()
        torch.cuda.synchronize()
        ms_copy, ms_inf = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
        print(f"H2D of the compressed bytes {ms_copy:.2f} ms, inflate {ms_inf:.2f} ms = {len(text) / ms_inf / 1e6:.1f} GB/s of text", flush=True)
    assert not bool(d_stat.any())
    for n_sub in (1, 64, 256, 1024, 2048, 4096, len(chunks)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _ffi.check(eng.lib.sai_inflate_bgzf(eng.ctx, C.c_void_p(d_comp.data_ptr()), d_comp.numel(), C.c_void_p(d_tab.data_ptr()),
                                            n_sub, C.c_void_p(d_text.data_ptr()), out + 8, C.c_void_p(d_stat.data_ptr()), None))  # fmt: skip
        e1.record()
        torch.cuda.synchronize()
        print(f"  first {n_sub:5d} members: {e0.elapsed_time(e1):.3f} ms", flush=True)
    got = d_text[:out].cpu().numpy().tobytes()
    assert got == text
    print("text identical; zlib.crc32 ok:", zlib.crc32(got) == zlib.crc32(text))
    h_text = torch.empty((out,), dtype=torch.uint8).pin_memory()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h_text.copy_(d_text[:out], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"D2H of the text into pinned memory: {1e3 * dt:.2f} ms = {out / dt / 1e9:.1f} GB/s")


if __name__ == "__main__":
    main()
