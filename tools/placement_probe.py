"""Does WHERE a block lies decide a few per cent of the site pass?  K candidate arrays of a C3 population's size
are allocated next to each other in one process, filled with the same synthetic genotypes, and the fused pass is
timed over every ordered (ref array, tgt array) pair of them through ONE scorer (``rebind``): a matrix of pass
times, the plain-read probe of every array, and the best and the worst pair once more at the end (does a pair keep
its time?).

    python tools/placement_probe.py [--arrays 8] [--passes 6] > gpurun_out/placement_probe.txt
"""

from __future__ import annotations

import argparse
import dataclasses
import sys
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--arrays", type=int, default=8)
    ap.add_argument("--passes", type=int, default=6)
    ap.add_argument("--workload", default="c3")
    a = ap.parse_args()
    import torch

    from sai_amd.engine import TiledPop

    dev = bench.HipDevice()
    dev.start(0)
    eng = dev.eng
    wl = bench.make_workload(a.workload)
    block, lay, _, scorer = dev.build(wl, 0, 1, SimpleNamespace(layout="int8", overlap="off"))
    torch.cuda.synchronize()
    ref, tgt = block.pops[0], block.pops[1]
    assert ref.n_ind == tgt.n_ind, "the matrix swaps arrays between the two roles"
    arrays = [ref.tiles, tgt.tiles]
    while len(arrays) < a.arrays:
        t = torch.empty_like(ref.tiles)
        t.copy_(arrays[len(arrays) % 2])  # array i holds the bytes of role i % 2
        arrays.append(t)
    torch.cuda.synchronize()
    print("arrays:", " ".join(f"{i}:{t.data_ptr():#x}" for i, t in enumerate(arrays)), flush=True)
    print("probe GB/s:", " ".join(f"{i}:{eng.probe_stream_read(t):.0f}" for i, t in enumerate(arrays)), flush=True)

    def timed(i: int, j: int, n: int) -> float:
        blk = dataclasses.replace(block, pops=[TiledPop(arrays[i], ref.n_sites, ref.n_ind), TiledPop(arrays[j], tgt.n_sites, tgt.n_ind),
                                               *block.pops[2:]])  # fmt: skip
        scorer.rebind(blk, wl.params())
        scorer.step()  # untimed
        before = len(scorer.site_pass_ms())
        for _ in range(n):
            scorer.step(time_counts=True)
        scorer.flush()
        torch.cuda.synchronize()
        ms = sorted(scorer.site_pass_ms()[before:])
        return ms[len(ms) // 2]

    K = len(arrays)
    t0 = time.perf_counter()
    m = [[float("nan")] * K for _ in range(K)]
    for i in range(K):
        for j in range(K):
            if i != j:
                m[i][j] = timed(i, j, a.passes)
    print(f"matrix of median pass ms (row = ref array, column = tgt array), {time.perf_counter() - t0:.1f} s:")
    print("      " + " ".join(f"{j:6d}" for j in range(K)))
    for i in range(K):
        print(f"{i:4d}  " + " ".join("   -  " if i == j else f"{m[i][j]:6.3f}" for j in range(K)), flush=True)
    cells = sorted((m[i][j], i, j) for i in range(K) for j in range(K) if i != j)
    print("best ", cells[:3], "worst", cells[-3:])
    row = [sum(m[i][j] for j in range(K) if j != i) / (K - 1) for i in range(K)]
    col = [sum(m[i][j] for i in range(K) if j != i) / (K - 1) for j in range(K)]
    print("mean as ref:", " ".join(f"{v:.3f}" for v in row))
    print("mean as tgt:", " ".join(f"{v:.3f}" for v in col))
    for tag, (_, i, j) in (("best", cells[0]), ("worst", cells[-1]), ("best", cells[0]), ("worst", cells[-1])):
        print(f"again {tag} ({i},{j}): {timed(i, j, 40):.3f} ms", flush=True)
    # what placement.py's own timer says about the same pairs (two populations, a set no site satisfies), and the same
    # with the block's source population in the pass
    from sai_amd import _ffi
    from sai_amd.placement import _PairTimer

    tm = _PairTimer(eng, ref.n_sites)
    pm = [[float("nan")] * K for _ in range(K)]
    p3 = [[float("nan")] * K for _ in range(K)]
    sets3 = [_ffi.make_params(0.0, 1.0, 0.5, [("=", 0.0)] * (len(block.pops) - 2), True)]

    def three(i, j):
        best = float("inf")
        for k in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.site_pass([TiledPop(arrays[i], ref.n_sites, ref.n_ind), TiledPop(arrays[j], tgt.n_sites, tgt.n_ind), *block.pops[2:]],
                          [1] * len(block.pops), sets3, out=tm.out, freq_mode="candidates")  # fmt: skip
            e1.record()
            e1.synchronize()
            if k:
                best = min(best, e0.elapsed_time(e1))
        return best

    for i in range(K):
        for j in range(K):
            if i != j:
                pm[i][j] = tm.ms(TiledPop(arrays[i], ref.n_sites, ref.n_ind), TiledPop(arrays[j], tgt.n_sites, tgt.n_ind))
                p3[i][j] = three(i, j)
    for name, mat in (("two-population timer", pm), ("timer with the source population", p3)):
        print(f"{name}, ms (and minus the scorer's pass):")
        for i in range(K):
            print(f"{i:4d}  " + " ".join("      -      " if i == j else f"{mat[i][j]:6.3f}({mat[i][j] - m[i][j]:+.3f})" for j in range(K)), flush=True)
    # is a pair slow everywhere?  Tenths of the sites of the worst and the best pair, slice k of ref with slice k of tgt
    # (the pass over views, as FeaturePreprocessor._score_in_parts takes them), then slice 0 of ref with every slice of tgt
    n_tiles = ref.n_sites // 64
    per = n_tiles // 10

    def view(i, k, n_ind):
        t = arrays[i][k * per * n_ind * 64 : (k + 1) * per * n_ind * 64]
        return TiledPop(t, per * 64, n_ind)

    for tag, (_, i, j) in (("worst", cells[-1]), ("best", cells[0])):
        tms = _PairTimer(eng, per * 64)
        gb = per * 64 * (ref.n_ind + tgt.n_ind) / 1e9
        same = [gb / tms.ms(view(i, k, ref.n_ind), view(j, k, tgt.n_ind), 3) * 1e3 for k in range(10)]
        cross = [gb / tms.ms(view(i, 0, ref.n_ind), view(j, k, tgt.n_ind), 3) * 1e3 for k in range(10)]
        print(f"{tag} pair ({i},{j}) in tenths, GB/s: slice k with slice k: " + " ".join(f"{v:.0f}" for v in same), flush=True)
        print(f"{tag} pair ({i},{j}) in tenths, GB/s: ref slice 0 with tgt slice k: " + " ".join(f"{v:.0f}" for v in cross), flush=True)
    # the product's answer (sai_amd/placement.py): the slowest pairs, settled
    from sai_amd.placement import settle_pair

    for _, i, j in cells[-4:] + cells[:2]:
        report = {}
        t0 = time.perf_counter()
        a2, b2 = settle_pair(eng, TiledPop(arrays[i], ref.n_sites, ref.n_ind), TiledPop(arrays[j], tgt.n_sites, tgt.n_ind), report=report)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        arrays += [a2.tiles, b2.tiles]
        print(f"settle ({i},{j}) matrix {m[i][j]:.3f} ms: {report['pairs'][0]} in {dt * 1e3:.0f} ms -> the scorer's pass "
              f"{timed(len(arrays) - 2, len(arrays) - 1, 20):.3f} ms", flush=True)  # fmt: skip
        del arrays[-2:], a2, b2
    print("probe GB/s again:", " ".join(f"{i}:{eng.probe_stream_read(t):.0f}" for i, t in enumerate(arrays)), flush=True)


if __name__ == "__main__":
    main()
