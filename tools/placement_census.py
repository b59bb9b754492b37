"""How many classes of allocations are there, and how big is each?  N arrays of a C3 population's size (zero-filled: the
class is a matter of where the bytes lie, not of what they are) are allocated one after the other until the card is
full; each is timed -- sai_amd/placement.py's two-population pass -- next to one representative of every class found so
far and joins the class it streams fast with, or founds a new one.  Prints the classes in allocation order.

    python tools/placement_census.py [--arrays 26] [--gb 10]
"""

from __future__ import annotations

import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--arrays", type=int, default=26)
    ap.add_argument("--n-ind", type=int, default=1000)
    ap.add_argument("--sites", type=int, default=10_000_000)
    a = ap.parse_args()
    import torch

    from sai_amd.engine import Engine, TiledPop
    from sai_amd.placement import _PairTimer

    eng = Engine.get(0)
    n_sites = (a.sites + 63) // 64 * 64
    nbytes = n_sites * a.n_ind
    timer = _PairTimer(eng, n_sites)
    pops = []
    for k in range(a.arrays):
        free, _ = torch.cuda.mem_get_info(eng.device)
        if free < nbytes + (2 << 30):
            print(f"card full after {k} arrays")
            break
        pops.append(TiledPop(torch.zeros((nbytes,), dtype=torch.int8, device=eng.device), n_sites, a.n_ind))
    torch.cuda.synchronize()
    # round by round: the first array without a class is the representative of a new one; every other array without a
    # class is timed next to it and joins when the pair streams at the fast level (known from the first round, in which
    # both levels show up); an array between the levels lies partly in the representative's class ("~")
    label = [None] * len(pops)
    fast_ms = None
    names = "ABCDEFGHIJ"
    n_cls = 0
    while any(v is None for v in label) and n_cls < len(names):
        todo = [k for k, v in enumerate(label) if v is None]
        rep, rest = todo[0], todo[1:]
        times = {k: timer.ms(pops[rep], pops[k]) for k in rest}
        if fast_ms is None:
            fast_ms = min(times.values())
        label[rep] = names[n_cls]
        part = []
        for k, t in times.items():
            if t <= fast_ms * 1.015:
                label[k] = names[n_cls]
            elif t <= fast_ms * 1.035:
                part.append(k)
        print(f"class {names[n_cls]}: representative array {rep}; ms next to it: " + " ".join(f"{k}:{t:.3f}" for k, t in times.items()), flush=True)
        if part:
            print(f"   between the levels (partly in class {names[n_cls]}): {part}")
        n_cls += 1
    print("classes in allocation order:", "".join(v or "?" for v in label))
    for c in names[:n_cls]:
        n = label.count(c)
        print(f"class {c}: {n} arrays = {n * nbytes / 1e9:.0f} GB")
    print("addresses:", " ".join(f"{k}:{p.tiles.data_ptr():#x}" for k, p in enumerate(pops)))


if __name__ == "__main__":
    main()
