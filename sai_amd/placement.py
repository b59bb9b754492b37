"""Where the big populations of a resident block lie relative to each other.

Measured on the pool's MI355X boxes (``profiles/r05_placement.txt``; ``tools/placement_probe.py``, ``placement_census.py``,
``placement_shift.py``, ``drift_probe.py``): how fast the site pass streams a (ref, tgt) pair depends on WHICH two
allocations hold them.  Every allocation of the card belongs to one of exactly THREE classes (27 arrays of 10 GB: 8 / 7 /
12, in runs of consecutive allocations); a pair of one class streams at the fast level (C3: 2.83-2.87 ms), every mixed
pair 5 % slower (2.96-3.02), in either order, in every tenth of its sites, for as long as it lives; an array that lies
partly in two classes sits in between.  The class belongs to where the allocation lies -- shifting an array inside it
by 4 KiB ... 1 GiB changes nothing, no bit of the virtual addresses tells it, temperature, clocks and power do not move
it, the plain-read rate of the single array is not in step with it -- so it cannot be asked for, but it can be measured,
and three classes make the search finite: of FOUR pieces of memory two are of one class.  ``settle_pair`` takes two
fresh pieces of memory next to the two arrays as built and times the pass over the six pairs of them that put the one
population into one piece and the other into another; the fastest pair is where the populations go.  Two pieces are
allocated and at most two populations copied, the bytes of the block do not change, only where they lie.  A block pays
eighteen passes once (C3: 80 ms next to an ingest of seconds); blocks under ``MIN_BYTES`` per population, and devices without
the room for another copy, are left as they are.

What the pass WRITES -- a row of flag planes per tile, the stored frequencies of the candidate sites -- is the other
way round (``tools/placement_outputs.py``): with the outputs inside a piece of the populations' own class C5's pass
(18 parameter sets, 55 MB written per launch) takes 3.10 ms, inside a piece of another class 2.97, and C3's (6 MB) 2.85
against 2.82-2.84.  The search knows such a piece -- a fresh one that made a slow pair with the chosen populations --
and hands a slice of it on as the block's ``OutputArena``; the scorer lays its per-site outputs there.

``SAI_AMD_PLACEMENT=0`` switches it off (A/B runs).
"""

from __future__ import annotations

import os
from typing import Optional, Sequence

from . import _ffi

MIN_BYTES = 1 << 30  # per population: below this a pass is too short for its placement to matter
GAIN = 0.01  # a placement must be faster than the present one by more than this to replace it (levels are 4-5 % apart, one level's spread is under 1 %)
MAX_TRIES = 3  # fresh pieces of memory tried for a population whose partner must stay where it is
PASSES = 2  # timed passes per pair (after one untimed); the faster counts
AWAY = 0.025  # a piece whose pair with a chosen population is slower than the chosen pair by more than this is of another class
ARENA_BYTES_PER_SITE = 48  # per-site outputs of a pipelined scorer with 20 parameter sets: 3 x (8 B stored frequency + 7.5 B of plane rows)
KEEP_FREE = 8 << 30  # bytes of HBM that stay free for the scorer's own buffers while rejected copies are held


def enabled() -> bool:
    return os.environ.get("SAI_AMD_PLACEMENT", "1") != "0"


def worth_moving(ms_now: float, ms_best: float, gain: float = GAIN) -> bool:
    """Is the best placement met faster than the present one by more than a level's own spread?"""
    return ms_best < ms_now * (1.0 - gain)


class OutputArena:
    """Memory of ANOTHER class than the block's big populations, for what the site pass writes.  ``take`` hands out
    views front to back and never takes one back: a scorer that finds it used up allocates as it always did."""

    def __init__(self, tensor):
        self.tensor, self.used = tensor, 0

    def take(self, nbytes: int, align: int = 256):
        start = (self.used + align - 1) // align * align
        if start + nbytes > self.tensor.numel():
            return None
        self.used = start + nbytes
        return self.tensor[start : start + nbytes]


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
                move_anchor: bool = True, arena: Optional[list] = None, arena_bytes: int = 0, hold: Optional[list] = None):  # fmt: skip
    """``(anchor, other)`` -- two TiledPop over the same sites, ``anchor`` the larger -- as they are, or copied into
    memory in which the pair streams faster.

    Candidates are pieces of memory: the two arrays as built and two fresh ones (``F1``, ``F2``, each large enough for
    ``anchor``).  All six pairs are timed -- as built; ``(anchor, F1)``, ``(F1, other)``, ``(anchor, F2)``, ``(F2, other)``:
    one copy to adopt; ``(F1, F2)``: two -- with the bytes of the role a piece plays copied into it first (what the
    bytes are decides which form of the stream loop runs).  With three classes of memory a pair of one class exists
    among four pieces; all six are timed rather than the first that looks fast, because an array that lies partly in
    two classes makes a pair in between the levels, which would pass for the fast one next to a slower pair.  The
    fastest pair is adopted when it beats the pair as built by more than ``GAIN`` (of two equally fast ones the one
    that costs fewer copies); everything else is released (the allocator's cache keeps the memory).
    ``move_anchor=False``: other populations have been settled next to this anchor already, it stays -- ``tries``
    fresh pieces are tried for ``other`` alone.
    ``arena`` (a list) receives an ``OutputArena`` of ``arena_bytes`` bytes when a fresh piece turned out to be of
    another class than the chosen pair: that piece is released first and the arena allocated right behind it, so that
    the allocator's cache carves it from there (checked by address; an arena that lies elsewhere is not handed on).
    ``hold`` (a list) receives the fresh pieces that are still alive at the end, so that the caller decides when the
    unused ones go back to the allocator."""
    from .engine import TiledPop

    import torch

    timer = timer or _PairTimer(eng, anchor.n_sites)
    log = {"n_ind": [anchor.n_ind, other.n_ind], "ms": [], "pairs": [], "moved": []}
    size_a, size_o = anchor.tiles.numel(), other.tiles.numel()
    fresh: list = []  # uint8/int8 tensors of max(size_a, size_o) bytes
    seen: dict = {}  # (anchor piece, other piece) -> ms; pieces: "a" / "o" = as built, 1, 2, ... = fresh[k - 1]

    def piece(name, role_pop, nbytes):
        if name in ("a", "o"):
            return role_pop
        view = fresh[name - 1][:nbytes]
        view.copy_(role_pop.tiles)
        return TiledPop(view, role_pop.n_sites, role_pop.n_ind)

    def measure(pa, po) -> None:
        seen[(pa, po)] = timer.ms(piece(pa, anchor, size_a), piece(po, other, size_o))
        log["ms"].append(round(seen[(pa, po)], 4))
        log["pairs"].append(f"{pa}{po}")

    def add_fresh() -> bool:
        need = max(size_a, size_o)
        if _free_bytes(eng) < need + KEEP_FREE:
            log["stopped"] = "no room for another copy"
            return False
        try:
            fresh.append(torch.empty((need,), dtype=anchor.tiles.dtype, device=anchor.tiles.device))
        except torch.cuda.OutOfMemoryError:  # another process took the room in between (several ranks on one card)
            log["stopped"] = "no room for another copy"
            return False
        return True

    measure("a", "o")
    if move_anchor:
        plan = [[("a", 1), (1, "o")], [("a", 2), (2, "o"), (1, 2)]]
    else:
        plan = [[("a", k)] for k in range(1, tries + 1)]
    for step in plan:
        if not add_fresh():
            break
        for pa, po in step:
            measure(pa, po)
    # the fastest pair; of pairs within a level's own spread of it, the one met first (fewest copies)
    fastest = min(seen.values())
    (pa, po), best = next(kv for kv in seen.items() if kv[1] <= fastest * (1.0 + GAIN / 2))
    if not worth_moving(seen[("a", "o")], best):
        pa, po, best = "a", "o", seen[("a", "o")]
    log["moved"] = [name for name, k in (("anchor", pa), ("other", po)) if k not in ("a", "o")]
    log["ms_chosen"] = round(best, 4)
    # an array a population has left is of another class only when ITS pair with the chosen partner was clearly slower
    # than the chosen pair (the test the fresh pieces go through below): a move for a per cent inside one class leaves
    # memory of the populations' own class behind -- the slow place for what the pass writes
    log["left_away"] = [name for name, k, old in (("anchor", pa, ("a", po)), ("other", po, (pa, "o")))
                        if k not in ("a", "o") and seen.get(old, 0.0) > best * (1.0 + AWAY)]  # fmt: skip
    if report is not None:
        report.setdefault("pairs", []).append(log)
    # the bytes go where they were chosen to lie (a fresh piece may have played the other role since)
    anchor2 = piece(pa, anchor, size_a)
    other2 = piece(po, other, size_o)
    if arena is not None and arena_bytes > 0:
        # a fresh piece that is in neither role and made a clearly slower pair with one of the chosen two: another class
        away = [k for k in range(1, len(fresh) + 1) if k not in (pa, po)
                and max(seen.get((pa, k), 0.0), seen.get((k, po), 0.0)) > best * (1.0 + AWAY)]  # fmt: skip
        if away and arena_bytes <= fresh[away[0] - 1].numel():
            lo = fresh[away[0] - 1].data_ptr()
            hi = lo + fresh[away[0] - 1].numel()
            fresh[away[0] - 1] = None  # back to the allocator's cache, alone: the next allocation that fits is carved from it
            got = _arena_from(eng, (lo, hi), arena_bytes, anchor.tiles.device)
            if got is not None:
                arena.append(got)
                log["arena"] = f"piece {away[0]}"
    if hold is not None:
        hold.extend(t for t in fresh if t is not None)
    return anchor2, other2


def _arena_from(eng, donor_range, nbytes: int, device):
    """An ``OutputArena`` of ``nbytes`` carved from memory that has just gone back to the allocator's cache
    (``donor_range`` = its [lo, hi) addresses), or None when the allocation landed elsewhere."""
    import torch

    lo, hi = donor_range
    if nbytes <= 0 or nbytes > hi - lo:
        return None
    elsewhere = []  # the cache hands out its smallest fitting block first: what lands elsewhere is held until the donor's turn
    try:
        for _ in range(8):
            got = torch.empty((nbytes,), dtype=torch.uint8, device=device)
            if lo <= got.data_ptr() and got.data_ptr() + nbytes <= hi:
                return OutputArena(got)
            elsewhere.append(got)
    except torch.cuda.OutOfMemoryError:
        pass
    return None


def settle_block(eng, pops: Sequence, report: Optional[dict] = None, arena: Optional[list] = None, owned: bool = False) -> list:
    """The populations of a block (TiledPop, in the block's order) with every big one settled next to the largest:
    see the module's docstring.  Returns the list to build the block from -- the same objects where nothing moved.
    ``arena`` (a list) receives the block's ``OutputArena`` when the search met memory of another class.
    ``owned=True``: ``pops`` is a list nobody else holds (nor its populations) -- it is changed in place, and a
    population that moved leaves its old array to the arena when that array made a clearly slower pair with the chosen
    partner -- it is of another class then -- and once nothing refers to it the allocator's cache hands its memory out
    again (an array left for a per cent inside one class is the populations' own class: no arena from it)."""
    if not owned:
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
    rep = report if report is not None else {}  # (settle_pair's log of a pair is read here too)
    timer = None
    anchor_is_fixed = False
    for i in big[1:]:
        if pops[i].n_sites != pops[anchor].n_sites:
            continue  # populations over other sites never stream in one pass
        if (pops[anchor].tiles.data_ptr(), pops[i].tiles.data_ptr(), size[i]) not in done:
            timer = timer or _PairTimer(eng, pops[anchor].n_sites)
            want = arena is not None and not arena and not anchor_is_fixed
            arena_bytes = ARENA_BYTES_PER_SITE * pops[anchor].n_sites
            hold: list = []  # the unused fresh pieces stay out of the allocator's cache until the arena has been carved
            new_a, new_o = settle_pair(eng, pops[anchor], pops[i], timer, report=rep, move_anchor=not anchor_is_fixed,
                                       arena=arena if want else None, arena_bytes=arena_bytes, hold=hold)  # fmt: skip
            left_away = rep["pairs"][-1]["left_away"]  # which of the arrays a population left are of another class (settle_pair)
            donors = [(p.tiles.data_ptr(), p.tiles.data_ptr() + p.tiles.numel())
                      for name, p, q in (("anchor", pops[anchor], new_a), ("other", pops[i], new_o)) if q is not p and name in left_away]  # fmt: skip
            device = pops[anchor].tiles.device
            pops[anchor], pops[i] = new_a, new_o
            del new_a, new_o
            if want and not arena and owned and donors:
                # nothing refers to the array a population has left any more (the caller's list was ours): its memory
                # is in the allocator's cache now, and the arena is the next allocation that fits
                got = _arena_from(eng, donors[0], arena_bytes, device)
                if got is not None:
                    arena.append(got)
                    if report is not None and report.get("pairs"):
                        report["pairs"][-1]["arena"] = "the array a population left"
            del hold
            done.add((pops[anchor].tiles.data_ptr(), pops[i].tiles.data_ptr(), size[i]))
        anchor_is_fixed = True
    return pops
