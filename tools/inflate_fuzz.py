"""One-off fuzz of sai_inflate_bgzf (GPU box): 18 000 members of mixed content, levels, strategies and
flushes, 30 % of them damaged; undamaged ones must come out exactly, damaged ones flagged or exact."""
import sys, zlib
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
from test_inflate_device import run, deflate, vcf_like
from sai_amd.engine import Engine
eng = Engine.get(0)
total = bad_total = 0
for seed in range(12):
    rng = np.random.default_rng(1000 + seed)
    base = vcf_like(rng, 1 << 20)
    streams, texts, corrupt = [], [], []
    for i in range(1500):
        kind = int(rng.integers(6))
        n = int(rng.integers(0, 65537))
        if kind == 0:
            t = bytes(rng.integers(0, 1 + int(rng.integers(1, 256)), size=n, dtype=np.uint8))
        elif kind == 1:
            p = bytes(rng.integers(0, 256, size=int(rng.integers(1, 40000)), dtype=np.uint8)); t = (p * (n // len(p) + 1))[:n]
        else:
            o = int(rng.integers(0, len(base) - n + 1)); t = base[o:o + n]
        s = bytearray(deflate(t, level=int(rng.integers(0, 10)), strategy=[zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY][int(rng.integers(5))], flush_every=[0, 0, 0, 5000, 300][int(rng.integers(5))]))
        c = rng.random() < 0.3 and len(s) > 4
        if c:
            for _ in range(int(rng.integers(1, 6))):
                s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(8))
            if rng.random() < 0.3:
                s = s[: int(rng.integers(1, len(s)))]
        streams.append(bytes(s)); texts.append(t); corrupt.append(c)
    status, outs, guards = run(eng, streams, texts, rng, int(rng.integers(0, 9)), int(rng.integers(0, 70)))
    assert guards, seed
    for i, (st, got, want, c) in enumerate(zip(status, outs, texts, corrupt)):
        if not c:
            assert st == 0 and got == want, (seed, i, int(st), len(want))
        else:
            assert st != 0 or got == want, (seed, i)
            bad_total += int(st != 0)
    total += len(streams)
    print("seed", seed, "ok", flush=True)
print(total, "members,", bad_total, "flagged")
