"""Where the big populations of a resident block lie relative to each other.

Measured on the pool's MI355X boxes (``tools/placement_probe.py``, ``profiles/r05_placement.txt``): how fast the site
pass streams a (ref, tgt) pair depends on WHICH two allocations hold them.  The arrays a process allocates fall into
a few classes (two or three among six to ten arrays); a pair of ONE class streams at the fast level (C3: 2.83-2.87 ms),
every mixed pair 5 % slower (2.96-3.01), in either order, and a pair keeps its level for as long as it lives.  The
class belongs to the allocation: shifting an array inside its allocation by 4 KiB ... 1 GiB changes nothing
(``tools/placement_shift.py``), no bit of the virtual addresses tells it, temperature, clocks and power do not move it
(``tools/drift_probe.py``), and the plain-read rate of a single array is not in step with it.  So it cannot be asked
for; it can be measured and chosen: ``settle_block`` times the pass over the anchor (the largest population) and each
other big population, then over fresh copies of that population, then over fresh copies of the anchor next to every
copy of the other -- until it meets a pair clearly of the fast kind (faster than the slowest pair met by more than
``LEVEL``) or runs out of tries -- and keeps the fastest pair when that is faster than the present one by more than
``GAIN``.  The bytes of the block do not change, only where they lie; a copy is one device copy.  A block pays a few
passes once (C3: 20-150 ms next to an ingest of seconds); blocks under ``MIN_BYTES`` per population, and devices
without the room for another copy, are left as they are.

``SAI_AMD_PLACEMENT=0`` switches it off (A/B runs).
"""

from __future__ import annotations

import os
from typing import Optional, Sequence

from . import _ffi

MIN_BYTES = 1 << 30  # per population: below this a pass is too short for its placement to matter
GAIN = 0.01  # a placement must be faster than the present one by more than this to replace it (levels are 4-5 % apart, one level's spread is under 1 %)
LEVEL = 0.03  # a pair faster than the slowest pair met by more than this is of the fast kind (the kinds are 5 % apart)
MAX_TRIES = 3  # fresh copies tried per population
PASSES = 2  # timed passes per pair (after one untimed); the faster counts
KEEP_FREE = 8 << 30  # bytes of HBM that stay free for the scorer's own buffers while rejected copies are held


def enabled() -> bool:
    return os.environ.get("SAI_AMD_PLACEMENT", "1") != "0"


def worth_moving(ms_now: float, ms_best: float, gain: float = GAIN) -> bool:
    """Is the best placement met faster than the present one by more than a level's own spread?"""
    return ms_best < ms_now * (1.0 - gain)


class _PairTimer:
    """The fused site pass over two populations with one parameter set no site satisfies (w = 0: nothing is
    written), timed with events on the current stream."""

    def __init__(self, eng, n_sites: int):
        import torch

        self.eng, self.torch = eng, torch
        self.sets = [_ffi.make_params(0.0, 1.0, 0.5, [], True)]
        self.out = (torch.empty((n_sites,), dtype=torch.float64, device=eng.device), eng.alloc_planes(n_sites, 1))

    def ms(self, a, b, passes: int = PASSES) -> float:
        torch = self.torch
        best = float("inf")
        for k in range(passes + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.eng.site_pass([a, b], [1, 1], self.sets, out=self.out, freq_mode="candidates")
            e1.record()
            e1.synchronize()
            if k:
                best = min(best, e0.elapsed_time(e1))
        return best


def _free_bytes(eng) -> int:
    import torch

    free, _total = torch.cuda.mem_get_info(eng.device)
    return int(free)


def settle_pair(eng, anchor, other, timer: Optional[_PairTimer] = None, tries: int = MAX_TRIES, report: Optional[dict] = None,
                move_anchor: bool = True):  # fmt: skip
    """``(anchor, other)`` -- two TiledPop over the same sites -- as they are, or copied into allocations in which the
    pair streams faster.

    The present pair is timed first, then up to ``tries`` fresh copies of ``other`` next to ``anchor``; if no pair
    clearly of the fast kind has shown up by then -- all the same: all fast, or all slow because ``anchor`` is of a
    kind of its own -- up to ``tries`` fresh copies of ``anchor``, each next to every copy of ``other`` held so far.
    All copies are held until the end, so that each lands elsewhere; the fastest pair met is kept when it beats the
    present one by more than ``GAIN``, the rest is released (the allocator's cache keeps the memory).
    ``move_anchor=False``: other populations have been settled next to this anchor already, it stays."""
    from .engine import TiledPop

    import torch

    timer = timer or _PairTimer(eng, anchor.n_sites)
    log = {"n_ind": [anchor.n_ind, other.n_ind], "ms": [], "moved": []}
    anchors, others, seen = [anchor], [other], {}

    def measure(ia: int, io: int) -> None:
        seen[(ia, io)] = timer.ms(anchors[ia], others[io])
        log["ms"].append(round(seen[(ia, io)], 4))

    def fast_kind_met() -> bool:
        return min(seen.values()) < max(seen.values()) * (1.0 - LEVEL)

    def add_copy(of: list) -> bool:
        pop = of[0]
        if _free_bytes(eng) < pop.tiles.numel() + KEEP_FREE:
            log["stopped"] = "no room for another copy"
            return False
        try:
            fresh = TiledPop(torch.empty_like(pop.tiles), pop.n_sites, pop.n_ind)
        except torch.cuda.OutOfMemoryError:  # another process took the room in between (several ranks on one card)
            log["stopped"] = "no room for another copy"
            return False
        fresh.tiles.copy_(pop.tiles)
        of.append(fresh)
        return True

    measure(0, 0)
    for _ in range(tries):
        if fast_kind_met() or not add_copy(others):
            break
        measure(0, len(others) - 1)
    for _ in range(tries if move_anchor else 0):
        if fast_kind_met() or "stopped" in log or not add_copy(anchors):
            break
        for io in range(len(others)):
            measure(len(anchors) - 1, io)
            if fast_kind_met():
                break
    (ia, io), best = min(seen.items(), key=lambda kv: kv[1])
    if not worth_moving(seen[(0, 0)], best):
        ia, io, best = 0, 0, seen[(0, 0)]
    log["moved"] = [name for name, k in (("anchor", ia), ("other", io)) if k]
    log["ms_chosen"] = round(best, 4)
    if report is not None:
        report.setdefault("pairs", []).append(log)
    return anchors[ia], others[io]


def settle_block(eng, pops: Sequence, report: Optional[dict] = None) -> list:
    """The populations of a block (TiledPop, in the block's order) with every big one settled next to the largest:
    see the module's docstring.  Returns the list to build the block from -- the same objects where nothing moved."""
    pops = list(pops)
    if report is not None:
        report["enabled"] = enabled()
    if not enabled() or len(pops) < 2:
        return pops
    size = [p.tiles.numel() for p in pops]
    big = sorted((i for i in range(len(pops)) if size[i] >= MIN_BYTES), key=lambda i: -size[i])
    if len(big) < 2:
        return pops
    anchor = big[0]
    # pairs this engine has settled already (a block handed from one generator to the next, the same region scored
    # again): known by where the two arrays lie -- memory that comes back from the allocator's cache is still where it was
    done = eng.__dict__.setdefault("_settled_pairs", set())
    timer = None
    anchor_is_fixed = False
    for i in big[1:]:
        if pops[i].n_sites != pops[anchor].n_sites:
            continue  # populations over other sites never stream in one pass
        if (pops[anchor].tiles.data_ptr(), pops[i].tiles.data_ptr(), size[i]) not in done:
            timer = timer or _PairTimer(eng, pops[anchor].n_sites)
            pops[anchor], pops[i] = settle_pair(eng, pops[anchor], pops[i], timer, report=report, move_anchor=not anchor_is_fixed)
            done.add((pops[anchor].tiles.data_ptr(), pops[i].tiles.data_ptr(), size[i]))
        anchor_is_fixed = True
    return pops
