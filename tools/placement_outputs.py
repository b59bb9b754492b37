"""Does it matter where the site pass WRITES?  C5's pass (18 parameter sets: 55 MB of plane rows and stored frequencies per
launch) over one settled block, with its outputs laid into pieces of memory whose class relative to the block's ref array
is known (placement.py's pair timer), and into a fresh allocation.

    python tools/placement_outputs.py [--workload c5] [--pieces 6]
"""

from __future__ import annotations

import argparse
import sys
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c5")
    ap.add_argument("--pieces", type=int, default=6)
    a = ap.parse_args()
    import torch

    from sai_amd import _ffi
    from sai_amd.engine import PLANES, TiledPop
    from sai_amd.placement import _PairTimer

    dev = bench.HipDevice()
    dev.start(0)
    eng = dev.eng
    wl = bench.make_workload(a.workload)
    block, lay, _, scorer = dev.build(wl, 0, 1, SimpleNamespace(layout="int8", overlap="off"))
    torch.cuda.synchronize()
    print("placement of the block:", block.extra.get("placement"), flush=True)
    ref, tgt = block.pops[0], block.pops[1]
    n, sets = ref.n_sites, wl.params()
    n_tiles = n // 64
    timer = _PairTimer(eng, n)
    print(f"pair as chosen: {timer.ms(ref, tgt):.3f} ms", flush=True)

    def pass_ms(out) -> float:
        ms = []
        for k in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.site_pass(block.pops, block.ploidies, sets, out=out, freq_mode="candidates")
            e1.record()
            e1.synchronize()
            if k:
                ms.append(e0.elapsed_time(e1))
        return sorted(ms)[len(ms) // 2]

    fresh = (torch.full((n,), float("nan"), dtype=torch.float64, device=eng.device), eng.alloc_planes(n, len(sets)))
    print(f"outputs in a fresh allocation ({fresh[0].data_ptr():#x}): {pass_ms(fresh):.3f} ms", flush=True)
    row = PLANES * len(sets)
    for k in range(a.pieces):
        piece = torch.zeros((ref.tiles.numel(),), dtype=torch.int8, device=eng.device)
        cls = timer.ms(ref, TiledPop(piece, n, ref.n_ind))
        freq = piece[: 8 * n].view(torch.float64)
        freq.fill_(float("nan"))
        planes = piece[8 * n : 8 * n + 8 * n_tiles * row].view(torch.int64).reshape(n_tiles, row)
        planes.zero_()
        print(f"piece {k} ({piece.data_ptr():#x}): next to ref as a population {cls:.3f} ms; the pass with its outputs inside it {pass_ms((freq, planes)):.3f} ms", flush=True)
        globals().setdefault("_keep", []).append(piece)  # every piece elsewhere
    print(f"outputs in the fresh allocation again: {pass_ms(fresh):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
