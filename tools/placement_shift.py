"""placement_probe.py found two classes of arrays: a (ref, tgt) pair of one class streams 5 % faster than a mixed
pair.  Here: ONE array for ref, and tgt laid at a growing offset inside a larger allocation -- which shifts of tgt's
start move the pass between its two levels?  Then the same with ref and tgt inside one arena.

    python tools/placement_shift.py > gpurun_out/placement_shift.txt
"""

from __future__ import annotations

import dataclasses
import sys
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402

MB = 1 << 20
SHIFTS = [0, 4096, 65536, MB, 2 * MB, 4 * MB, 8 * MB, 16 * MB, 32 * MB, 48 * MB, 64 * MB, 96 * MB, 128 * MB, 160 * MB, 192 * MB,
          224 * MB, 256 * MB, 320 * MB, 384 * MB, 448 * MB, 512 * MB, 576 * MB, 640 * MB, 768 * MB, 896 * MB, 1024 * MB]  # fmt: skip


def main() -> None:
    import torch

    from sai_amd.engine import TiledPop

    dev = bench.HipDevice()
    dev.start(0)
    wl = bench.make_workload("c3")
    block, lay, _, scorer = dev.build(wl, 0, 1, SimpleNamespace(layout="int8", overlap="off"))
    torch.cuda.synchronize()
    ref, tgt = block.pops[0], block.pops[1]
    n = ref.tiles.numel()

    def timed(r, t, passes=6) -> float:
        blk = dataclasses.replace(block, pops=[TiledPop(r, ref.n_sites, ref.n_ind), TiledPop(t, tgt.n_sites, tgt.n_ind), *block.pops[2:]])
        scorer.rebind(blk, wl.params())
        scorer.step()
        before = len(scorer.site_pass_ms())
        for _ in range(passes):
            scorer.step(time_counts=True)
        scorer.flush()
        torch.cuda.synchronize()
        ms = sorted(scorer.site_pass_ms()[before:])
        return ms[len(ms) // 2]

    print(f"as built: ref {ref.tiles.data_ptr():#x} tgt {tgt.tiles.data_ptr():#x}: {timed(ref.tiles, tgt.tiles):.3f} ms", flush=True)
    slack = SHIFTS[-1]
    for trial in range(2):
        big = torch.empty((n + slack,), dtype=torch.int8, device=dev.device)
        print(f"tgt inside its own allocation {big.data_ptr():#x} (trial {trial}), ref as built:", flush=True)
        for s in SHIFTS:
            view = big[s : s + n]
            view.copy_(tgt.tiles)
            print(f"  shift {s / MB:9.3f} MB: {timed(ref.tiles, view):.3f} ms", flush=True)
        del view, big
    arena = torch.empty((2 * n + slack + 4 * MB,), dtype=torch.int8, device=dev.device)
    base = (n + 2 * MB - 1) // (2 * MB) * (2 * MB)
    r = arena[:n]
    r.copy_(ref.tiles)
    print(f"ref and tgt in one arena {arena.data_ptr():#x}, tgt at {base / MB:.0f} MB + shift:", flush=True)
    for s in SHIFTS:
        view = arena[base + s : base + s + n]
        view.copy_(tgt.tiles)
        print(f"  shift {s / MB:9.3f} MB: {timed(r, view):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
