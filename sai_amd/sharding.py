"""Whole-genome window sharding: which windows a rank owns and which sites it must hold.

The reference cuts ONE chromosome's window list into contiguous ranges, one per worker
(sai/generators/chunk_generator.py:111-142; the executor sai/multiprocessing/mp_pool.py:45-73 hands
each range to a process that re-reads its own region).  The multi-GPU build applies the same rule to
the window list of the whole job -- the chromosomes' lists laid end to end -- so 22 chromosomes
divide evenly over 8 GPUs: rank r owns windows [g0, g1) of the global list (the first
``n % world`` ranks own one more), i.e. a tail of one chromosome, some whole chromosomes and a head
of another.  Each such *piece* needs the sites from its first window's start to its last window's
end -- its own range plus a halo of ``win_len - win_step`` bp shared with the neighbouring rank,
which is loaded or generated again, never exchanged.  The pieces of a rank are laid back to back,
tile-aligned, in ONE resident block (sai_amd.resident.ResidentBlock.segments), so a pass is still
one genotype stream and one windows stage whatever the number of pieces.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _ffi
from .utils.windows import split_genome, split_index_ranges

TILE = _ffi.SAI_TILE_SITES


@dataclass(frozen=True)
class Piece:
    """Windows [w0, w1) of chromosome ``chrom_index``'s own window list; ``g0`` = position of
    window w0 in the job's global list."""

    chrom_index: int
    w0: int
    w1: int
    g0: int

    @property
    def n_windows(self) -> int:
        return self.w1 - self.w0


def plan_shards(windows_per_chrom: Sequence[int], world: int) -> list[list[Piece]]:
    """Pieces of every rank (a rank beyond the number of windows gets none)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    total = int(sum(windows_per_chrom))
    ranges = split_index_ranges(total, world) if total else []
    first = np.concatenate([[0], np.cumsum(np.asarray(windows_per_chrom, dtype=np.int64))])
    plan: list[list[Piece]] = [[] for _ in range(world)]
    for rank, (g0, g1) in enumerate(ranges):
        c = int(np.searchsorted(first, g0, side="right")) - 1
        g = g0
        while g < g1:
            while first[c + 1] <= g:  # chromosomes without windows
                c += 1
            end = min(g1, int(first[c + 1]))
            plan[rank].append(Piece(c, g - int(first[c]), end - int(first[c]), g))
            g = end
    return plan


@dataclass
class SynthWorkload:
    """A synthetic job in the terms of SURVEY.md section 8(d): chromosome ids, sizes, the window
    grid and the parameter sets answered from one genotype pass."""

    name: str
    seed: int
    chroms: list
    n_sites: int  # per chromosome
    n_ref: int
    n_tgt: int
    src_sizes: list
    win_len: int
    win_step: int
    specs: list  # dicts w, x, quantile, y_list, anc
    ploidy: int = 2
    missing_per_million: int = 0
    description: str = ""

    @property
    def pop_sizes(self) -> list:
        return [self.n_ref, self.n_tgt, *self.src_sizes]

    def params(self) -> list:
        return [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in self.specs]


@dataclass
class ShardLayout:
    """Where a rank's pieces lie inside its block (host-side bookkeeping, no device memory)."""

    pieces: list  # Piece
    site0: list  # first site of the chromosome each piece holds
    n_sites: list  # sites per piece
    tile0: list  # first tile of each piece inside the block
    windows: list = field(default_factory=list)  # (chrom id, start, end) per owned window, global order
    window_segment: Optional[np.ndarray] = None

    @property
    def n_tiles(self) -> int:
        return sum((n + TILE - 1) // TILE for n in self.n_sites)

    @property
    def segments(self) -> list:
        return [(t * TILE, t * TILE + n) for t, n in zip(self.tile0, self.n_sites)]


def piece_site_range(pos: np.ndarray, windows: Sequence[tuple], w0: int, w1: int) -> tuple[int, int]:
    """[site0, site1) a piece needs: from its first window's start to its last window's end
    (window_generator.py:173-183's inclusive bounds)."""
    a = int(np.searchsorted(pos, windows[w0][0], side="left"))
    b = int(np.searchsorted(pos, windows[w1 - 1][1], side="right"))
    return a, max(a, b)


def layout_shard(pieces: Sequence[Piece], chrom_ids: Sequence, chrom_windows: Sequence[Sequence[tuple]],
                 site_range) -> ShardLayout:  # fmt: skip
    """``site_range(piece) -> (site0, site1)`` of the chromosome's site list."""
    site0, n_sites, tile0, windows, wseg = [], [], [], [], []
    t = 0
    for k, pc in enumerate(pieces):
        a, b = site_range(pc)
        site0.append(a)
        n_sites.append(b - a)
        tile0.append(t)
        t += (b - a + TILE - 1) // TILE
        for w in range(pc.w0, pc.w1):
            s, e = chrom_windows[pc.chrom_index][w]
            windows.append((chrom_ids[pc.chrom_index], s, e))
            wseg.append(k)
    return ShardLayout(list(pieces), site0, n_sites, tile0, windows, np.asarray(wseg, dtype=np.int64))


def synth_chrom_windows(lib, wl: SynthWorkload) -> tuple[list, list]:
    """(positions per chromosome as host int32 arrays, window list per chromosome).  Host side (the
    counter-based gap generator of the library), identical on every rank."""
    import ctypes as C

    all_pos, all_windows = [], []
    for chrom in wl.chroms:
        gaps = np.empty(wl.n_sites, dtype=np.int32)
        _ffi.check(lib.sai_synth_gaps_host(wl.seed, int(chrom), 0, wl.n_sites, gaps.ctypes.data_as(C.c_void_p)))
        pos64 = np.cumsum(gaps, dtype=np.int64)
        if wl.n_sites and pos64[-1] >= 2**31:
            raise ValueError("synthetic chromosome exceeds int32 coordinates")
        pos = pos64.astype(np.int32)
        all_pos.append(pos)
        all_windows.append(split_genome([int(pos[0]), int(pos[-1])], wl.win_len, wl.win_step) if wl.n_sites else [])
    return all_pos, all_windows


def build_synth_shard(eng, wl: SynthWorkload, rank: int, world: int):
    """This rank's share of a synthetic job, generated in place in HBM: returns
    ``(ResidentBlock, ShardLayout, windows_per_chrom)``; ``block`` is None when the rank owns no
    window.  Every byte is a pure function of (seed, chromosome, site, population, individual), so
    the shard holds exactly the bytes a single-GPU run holds at those sites."""
    import torch

    from .resident import ResidentBlock

    all_pos, all_windows = synth_chrom_windows(eng.lib, wl)
    counts = [len(w) for w in all_windows]
    pieces = plan_shards(counts, world)[rank]
    lay = layout_shard(pieces, wl.chroms, all_windows,
                       lambda pc: piece_site_range(all_pos[pc.chrom_index], all_windows[pc.chrom_index], pc.w0, pc.w1))  # fmt: skip
    if not pieces:
        return None, lay, counts
    n_tiles = lay.n_tiles
    from .engine import TiledPop

    pops = []
    for stream, n_ind in enumerate(wl.pop_sizes):
        tiles = torch.empty((n_tiles * n_ind * TILE,), dtype=torch.int8, device=eng.device)
        for pc, a, n, t0 in zip(lay.pieces, lay.site0, lay.n_sites, lay.tile0):
            nt = (n + TILE - 1) // TILE
            eng.synth_population(wl.seed, int(wl.chroms[pc.chrom_index]), a, n, stream, n_ind, wl.ploidy,
                                 wl.missing_per_million, out=tiles[t0 * n_ind * TILE : (t0 + nt) * n_ind * TILE])  # fmt: skip
        pops.append(TiledPop(tiles, n_tiles * TILE, n_ind))
    from .placement import settle_block

    placement: dict = {}
    arena: list = []
    pops = settle_block(eng, pops, placement, arena, owned=True)  # the same bytes, possibly in another allocation (placement.py)
    pos_host = np.zeros(n_tiles * TILE, dtype=np.int32)
    for pc, a, n, t0 in zip(lay.pieces, lay.site0, lay.n_sites, lay.tile0):
        pos_host[t0 * TILE : t0 * TILE + n] = all_pos[pc.chrom_index][a : a + n]
    pos = torch.from_numpy(pos_host).to(eng.device)
    block = ResidentBlock(pops, [wl.ploidy] * len(pops), pos, segments=lay.segments, extra={"placement": placement, "output_arena": arena[0] if arena else None})
    return block, lay, counts


def merge_rank_results(per_rank: Sequence, plan: Sequence[Sequence[Piece]], n_sets: int):
    """Rank-ordered WindowResults (None for ranks without windows) -> one WindowResults over the
    global window list: records [set][global window], candidate lists re-laid in (set, window)
    order -- exactly what a single-GPU run of the same job returns."""
    from .engine import RECORD_DTYPE, WindowResults

    total = sum(pc.n_windows for pieces in plan for pc in pieces)
    rec = np.zeros((n_sets, total), dtype=RECORD_DTYPE)
    u_parts, q_parts = [], []
    for s in range(n_sets):
        for res, pieces in zip(per_rank, plan):
            if not pieces:
                continue
            g0 = pieces[0].g0
            n_w = sum(pc.n_windows for pc in pieces)
            if res is None or res.records.shape != (n_sets, n_w):
                raise ValueError("a rank's records do not match the shard plan")
            if s == 0 and (int(res.records["u_count"].sum()) != res.cdd_u.size or int(res.records["n_cdd_q"].sum()) != res.cdd_q.size):
                raise ValueError("a rank's candidate lists do not match its records (overflowed or truncated row)")
            rec[s, g0 : g0 + n_w] = res.records[s]
            if n_w:
                a = int(res.offsets[s, 0, 0])
                b = int(res.offsets[s, n_w - 1, 0]) + int(res.records[s, n_w - 1]["u_count"])
                u_parts.append(res.cdd_u[a:b])
                a = int(res.offsets[s, 0, 1])
                b = int(res.offsets[s, n_w - 1, 1]) + int(res.records[s, n_w - 1]["n_cdd_q"])
                q_parts.append(res.cdd_q[a:b])
    off = np.zeros((n_sets, total, 2), dtype=np.int64)
    for k, name in enumerate(("u_count", "n_cdd_q")):
        flat = rec[name].reshape(-1).astype(np.int64)
        off[:, :, k] = (np.cumsum(flat) - flat).reshape(n_sets, total)
    cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0, dtype=np.int32)  # noqa: E731
    return WindowResults(rec, off, cat(u_parts), cat(q_parts))
