"""Resident chromosome region + its window grid (mirror of
sai/generators/window_generator.py:28-318).

The reference materialises every window as freshly sliced ``[sites][individuals]`` matrices
with full-chromosome masks (O(N_total) per window).  Here the region stays resident once
(int8 dosages) and a window is a site-index range found by binary search; ``get()`` still
yields the reference's 13-key dictionaries (the matrices are views) for user code and for the
single-window statistic API, while the batched GPU path (FeaturePreprocessor.run_windows)
uploads the resident blocks once and never builds per-window matrices.
"""

from __future__ import annotations

from itertools import combinations, product
from typing import Any, Iterator, Optional

import numpy as np

from ..utils.genomic_dataclasses import ChromosomeData
from ..utils.read_data import read_dosage_data
from ..utils.windows import split_genome
from .data_generator import DataGenerator


class AlignedSites:
    """Row selection of one population combination (``WindowGenerator.aligned``)."""

    __slots__ = ("keys", "pos_rows", "uniq", "rows", "file_order", "segments")

    def __init__(self, keys, pos_rows, uniq, rows, file_order=None, segments=None):
        self.keys = keys  # [(group, population)] in ref, tgt, sources, outgroup order
        self.pos_rows = pos_rows  # position of every selected row, non-decreasing
        self.uniq = uniq  # the unique common positions when rows repeat a position, else None
        self.rows = rows  # {(group, population): selected row indices}, or None when every row is kept
        # populations that share ONE position array which does not ascend (an unsorted VCF): the rows are
        # gathered in position order (`rows`), and file_order[j] = index in the file of sorted row j -- the
        # reference's matrices keep the file order while its `pos` is sorted (window_generator.py:193-231)
        self.file_order = file_order
        # populations whose position arrays do not ascend AND differ: the reference pairs, window by window, row k
        # of one population's selection (file order) with row k of every other's and with the k-th smallest
        # common position, so no single row order serves all windows -- the selected rows are the windows'
        # own selections laid end to end: segments[w] = [lo, hi) of window w in `rows` / `pos_rows`
        self.segments = segments


class WindowGenerator(DataGenerator):
    def __init__(
        self,
        vcf_file: str,
        chr_name: str,
        ref_ind_file: str,
        tgt_ind_file: str,
        src_ind_file: str,
        out_ind_file: Optional[str],
        win_len: int,
        win_step: int,
        ploidy_config,
        start: int = None,
        end: int = None,
        anc_allele_file: str = None,
        num_src: int = 1,
        resident: bool = False,
        preloaded=None,
    ):
        """``resident=True`` (the batched GPU driver, ChunkPreprocessor): the region is streamed to
        the GPU and tokenised there, the populations exist only as tiled blocks in HBM and ``get()``
        is not available; otherwise the populations are host matrices, as in the reference.
        ``preloaded`` = what ``read_data_device`` returned for exactly this region (the caller read it
        before it knew the chunk bounds, ChunkPreprocessor.preload)."""
        if win_len <= 0:
            raise ValueError("`win_len` must be greater than 0.")
        if win_step < 0:
            raise ValueError("`win_step` must be non-negative.")
        if num_src < 1:
            raise ValueError("`num_src` must be at least 1.")
        kw = dict(vcf_file=vcf_file, chr_name=chr_name, start=start, end=end, ref_ind_file=ref_ind_file,
                  tgt_ind_file=tgt_ind_file, src_ind_file=src_ind_file, out_ind_file=out_ind_file,
                  ploidy_config=ploidy_config, anc_allele_file=anc_allele_file)  # fmt: skip
        pos_dev = None
        if preloaded is not None:
            results, pos_dev = preloaded
        elif resident:
            from ..engine import Engine
            from ..utils.read_data import read_data_device

            results, pos_dev = read_data_device(Engine.get(), **kw)
        else:
            results = read_dosage_data(**kw)  # = read_data(..., is_phased=False, no filters): window_generator.py:105-120
        self._setup(
            chr_name, win_len, win_step, ploidy_config, results["ref"], results["tgt"], results["src"], start, end, num_src,
            results["outgroup"],
        )
        if pos_dev is not None:
            first = next(iter(self.ref_data.values())) if self.ref_data else None
            if first is not None:
                self._device_pos = (first.POS, pos_dev)

    @classmethod
    def from_arrays(
        cls,
        chr_name,
        ref_data: dict,
        tgt_data: dict,
        src_data: dict,
        win_len: int,
        win_step: int,
        ploidy_config,
        start: int = None,
        end: int = None,
        num_src: Optional[int] = None,
        out_data: Optional[dict] = None,
    ) -> "WindowGenerator":
        """Build from in-memory ``{population: ChromosomeData}`` dictionaries (synthetic data,
        tests); sample names are synthesised."""
        self = object.__new__(cls)

        def names(d):
            return {k: [f"{k}_{i}" for i in range(v.GT.shape[1])] for k, v in d.items()}

        self._setup(
            chr_name, win_len, win_step, ploidy_config, (ref_data, names(ref_data)), (tgt_data, names(tgt_data)),
            (src_data, names(src_data)), start, end, len(src_data) if num_src is None else num_src,
            (out_data, names(out_data)) if out_data else (None, None),
        )  # fmt: skip
        return self

    @classmethod
    def from_resident(cls, chr_name, pos: np.ndarray, pos_dev, ref: dict, tgt: dict, src: dict, win_len: int,
                      win_step: int, ploidy_config, start: int = None, end: int = None, out: Optional[dict] = None):  # fmt: skip
        """Build from blocks that already live in HBM: ``ref`` / ``tgt`` / ``src`` / ``out`` map
        population -> TiledPop over the sites of ``pos`` (host int32 array; ``pos_dev`` its device
        copy).  ``get()`` (per-window host matrices for the plugin classes) is not available on such
        a generator; the batched path (FeaturePreprocessor.run_windows) is."""
        data = lambda d: {k: ChromosomeData(POS=pos, REF=None, ALT=None, GT=v) for k, v in d.items()}  # noqa: E731
        names = lambda d: {k: [f"{k}_{i}" for i in range(v.n_ind)] for k, v in d.items()}  # noqa: E731
        self = object.__new__(cls)
        self._setup(chr_name, win_len, win_step, ploidy_config, (data(ref), names(ref)), (data(tgt), names(tgt)),
                    (data(src), names(src)), start, end, len(src), (data(out), names(out)) if out else (None, None))  # fmt: skip
        self._device_pos = (pos, pos_dev)
        return self

    def _setup(self, chr_name, win_len, win_step, ploidy_config, ref, tgt, src, start, end, num_src, out=(None, None)):
        self.win_len, self.win_step, self.num_src = win_len, win_step, num_src
        self.chr_name, self.ploidy_config = chr_name, ploidy_config
        self.start, self.end = start, end
        self.ref_data, self.ref_samples = ref
        self.tgt_data, self.tgt_samples = tgt
        self.src_data, self.src_samples = src
        self.out_data, self.out_samples = out
        # window_generator.py:129-131: combinations of the populations of the source file
        self.src_combinations = list(combinations(self.src_samples.keys(), self.num_src))
        # :132-144: the grid comes from the target's positions, or from the chunk bounds
        self.tgt_windows = {}
        for tgt_pop in self.tgt_samples:
            if start is None and end is None:
                if self.tgt_data is None or tgt_pop not in self.tgt_data:
                    raise ValueError(f"no variant data for target population '{tgt_pop}'")
                grid_pos = self.tgt_data[tgt_pop].POS
            else:
                grid_pos = [start, end - win_len + win_step]
            self.tgt_windows[tgt_pop] = split_genome(grid_pos, win_len, win_step, start=start)
        self.total_windows = sum(
            len(w) * len(self.ref_samples) * len(self.src_combinations) for w in self.tgt_windows.values()
        )

    # -- shared by the compat generator and the batched GPU path ---------------------------

    def has_data(self) -> bool:
        return self.ref_data is not None and self.tgt_data is not None and self.src_data is not None

    def combinations(self) -> Iterator[tuple]:
        """(ref_pop, tgt_pop, src_comb, out_pop) in the reference's product order (:162-166)."""
        return product(self.ref_samples, self.tgt_samples, self.src_combinations, self.out_samples or [None])

    def aligned(self, ref_pop, tgt_pop, src_comb, out_pop=None) -> "AlignedSites":
        """The rows of every population of one combination that the reference would select
        (window_generator.py:193-231: per window, ``intersect1d`` over the populations' positions,
        then ``isin`` per population).  Selecting by position set commutes with cutting a window
        out, so the intersection is taken ONCE for the whole region and a window is a row range.

        Usual case -- all populations come from one VCF region: identical, strictly increasing
        positions, nothing to gather.  Otherwise (populations lacking sites, repeated positions)
        each population gets the index array of its selected rows; positions must be sorted."""
        keys = [("ref", ref_pop), ("tgt", tgt_pop)] + [("src", s) for s in src_comb]
        if out_pop is not None:
            keys.append(("outgroup", out_pop))
        cache = self.__dict__.setdefault("_aligned", {})
        if tuple(keys) in cache:
            return cache[tuple(keys)]
        group_data = {"ref": self.ref_data, "tgt": self.tgt_data, "src": self.src_data, "outgroup": self.out_data}
        blocks = [group_data[g][p] for g, p in keys]
        pos = blocks[0].POS
        same = all(b.POS is pos or (b.POS.shape == pos.shape and np.array_equal(b.POS, pos)) for b in blocks[1:])
        if same and (pos.size < 2 or np.all(pos[1:] > pos[:-1])):
            out = AlignedSites(keys, pos, None, None)
        elif same and np.unique(pos).size == pos.size:
            # one unsorted position array for all: work on the rows in position order, remember the file order
            order = np.argsort(pos, kind="stable")
            out = AlignedSites(keys, pos[order], None, {k: order for k in keys}, file_order=order)
        else:
            if any(b.POS.size > 1 and not np.all(b.POS[1:] >= b.POS[:-1]) for b in blocks):
                out = self._aligned_per_window(keys, blocks, tgt_pop)
                cache[tuple(keys)] = out
                return out
            common = np.unique(pos)
            for b in blocks[1:]:
                common = np.intersect1d(common, b.POS)
            rows = [np.flatnonzero(np.isin(b.POS, common)) for b in blocks]
            row_pos = [b.POS[r] for b, r in zip(blocks, rows)]
            for rp in row_pos[1:]:
                if rp.shape != row_pos[0].shape or not np.array_equal(rp, row_pos[0]):
                    # a position repeated in some populations only: the reference's matrices of a window
                    # then have different numbers of rows and numpy refuses to combine them
                    raise ValueError(
                        f"operands could not be broadcast together with shapes ({row_pos[0].size},) ({rp.size},) "
                    )
            identity = all(r.size == b.POS.size for r, b in zip(rows, blocks))
            uniq = common if common.size != row_pos[0].size else None
            out = AlignedSites(keys, row_pos[0], uniq, None if identity else dict(zip(keys, rows)))
        cache[tuple(keys)] = out
        return out

    def _aligned_per_window(self, keys, blocks, tgt_pop) -> "AlignedSites":
        """window_generator.py:173-231 taken literally, for position arrays that are unsorted and differ: per
        window the common positions (sorted, unique) and, per population, the rows that carry them IN FILE
        ORDER.  A position repeated inside a population has no batched form here."""
        if any(np.unique(b.POS).size != b.POS.size for b in blocks):
            raise NotImplementedError("populations with different unsorted position arrays that repeat a position")
        common = np.unique(blocks[0].POS)
        for b in blocks[1:]:
            common = np.intersect1d(common, b.POS)
        order = [np.argsort(b.POS, kind="stable") for b in blocks]
        sorted_pos = [b.POS[o] for b, o in zip(blocks, order)]
        rows, pos_parts, segments, at = [[] for _ in blocks], [], [], 0
        for start, end in self.tgt_windows[tgt_pop]:
            here = common[slice(*self.window_range(common, start, end))]
            for k in range(len(blocks)):
                rows[k].append(np.sort(order[k][np.searchsorted(sorted_pos[k], here)]))
            pos_parts.append(here)
            segments.append((at, at + here.size))
            at += here.size
        cat = lambda parts, dt: np.concatenate(parts).astype(dt, copy=False) if parts else np.zeros(0, dtype=dt)  # noqa: E731
        return AlignedSites(keys, cat(pos_parts, blocks[0].POS.dtype), None, {k: cat(r, np.int64) for k, r in zip(keys, rows)},
                            segments=segments)  # fmt: skip

    def common_positions(self, ref_pop, tgt_pop, src_comb, out_pop=None) -> np.ndarray:
        """Positions of the rows shared by the populations of one combination."""
        return self.aligned(ref_pop, tgt_pop, src_comb, out_pop).pos_rows

    def device_blocks(self, eng) -> dict:
        """{(group, population): TiledPop} of every loaded population: host matrices are uploaded
        and re-tiled once per generator; blocks that already live in HBM (``from_resident``) are
        handed through."""
        from ..engine import TiledPop

        cache = self.__dict__.setdefault("_device_blocks", {})
        fresh = False
        groups = [("ref", self.ref_data), ("tgt", self.tgt_data), ("src", self.src_data)]
        if self.out_data:
            groups.append(("outgroup", self.out_data))
        for group, data in groups:
            for pop, cd in data.items():
                if (group, pop) not in cache:
                    cache[(group, pop)] = cd.GT if isinstance(cd.GT, TiledPop) else eng.tile(cd.GT)
                    fresh = True
        if fresh:  # big populations are settled next to the largest one once (placement.py): same bytes, maybe elsewhere
            from ..placement import settle_block

            keys = list(cache)
            owner = {(group, pop): cd for group, data in groups for pop, cd in data.items()}
            arena: list = []
            settled = settle_block(eng, [cache[k] for k in keys], arena=arena)
            if arena:  # memory of another class than the populations': where the scorers of this region write (placement.py)
                self.__dict__["_output_arena"] = arena[0]
            for key, pop in zip(keys, settled):
                if pop is not cache[key]:
                    cache[key] = pop
                    if isinstance(owner[key].GT, TiledPop):  # a resident population: the moved copy is the population now
                        owner[key].GT = pop
        return cache

    def device_positions(self, eng, pos: np.ndarray):
        """int32 device tensor of a position array (resident blocks bring their own)."""
        import torch

        dev = self.__dict__.get("_device_pos")
        if dev is not None and dev[0] is pos:
            return dev[1]
        return torch.as_tensor(np.ascontiguousarray(pos, dtype=np.int32)).to(eng.device)

    def __getstate__(self):  # device handles never travel with a pickled generator
        return {k: v for k, v in self.__dict__.items() if k not in ("_device_blocks", "_device_pos", "_aligned", "_scorers", "_win_arrays", "_part_plans", "_output_arena")}

    @staticmethod
    def window_range(pos: np.ndarray, start: int, end: int) -> tuple[int, int]:
        """[lo, hi) site indices of the inclusive window (window_generator.py:173-183)."""
        return int(np.searchsorted(pos, start, side="left")), int(np.searchsorted(pos, end, side="right"))

    # -- reference-compatible iteration ----------------------------------------------------

    def _empty_item(self, ref_pop, tgt_pop, src_comb, out_pop, start, end) -> dict[str, Any]:
        return {
            "chr_name": self.chr_name, "ref_pop": ref_pop, "tgt_pop": tgt_pop, "src_pop_list": src_comb,
            "out_pop": out_pop, "start": start, "end": end, "pos": [], "ref_gts": None, "tgt_gts": None,
            "src_gts_list": None, "out_gts": None, "ploidy_config": self.ploidy_config,
        }  # fmt: skip

    def _window_generator(self) -> Iterator[dict[str, Any]]:
        if "_device_pos" in self.__dict__:
            raise TypeError("a generator over HBM-resident blocks has no per-window host matrices; use run_windows")
        group_data = {"ref": self.ref_data, "tgt": self.tgt_data, "src": self.src_data, "outgroup": self.out_data}
        for ref_pop, tgt_pop, src_comb, out_pop in self.combinations():
            al = self.aligned(ref_pop, tgt_pop, src_comb, out_pop)
            mats = {}
            for key in al.keys:
                gt = group_data[key[0]][key[1]].GT
                mats[key] = gt if al.rows is None else gt[al.rows[key]]
            for wi, (start, end) in enumerate(self.tgt_windows[tgt_pop]):
                lo, hi = al.segments[wi] if al.segments is not None else self.window_range(al.pos_rows, start, end)
                if hi <= lo:  # window_generator.py:199-215
                    yield self._empty_item(ref_pop, tgt_pop, src_comb, out_pop, start, end)
                    continue
                item = self._empty_item(ref_pop, tgt_pop, src_comb, out_pop, start, end)
                if al.file_order is not None:  # the window's rows in FILE order next to the sorted positions
                    in_file = np.sort(al.file_order[lo:hi])
                    full = {key: group_data[key[0]][key[1]].GT for key in al.keys}
                    item.update(
                        pos=al.pos_rows[lo:hi], ref_gts=full[("ref", ref_pop)][in_file], tgt_gts=full[("tgt", tgt_pop)][in_file],
                        src_gts_list=[full[("src", s)][in_file] for s in src_comb],
                        out_gts=None if out_pop is None else full[("outgroup", out_pop)][in_file],
                    )  # fmt: skip
                    yield item
                    continue
                item.update(
                    # the reference hands over the UNIQUE common positions next to matrices that keep
                    # every row of a repeated position (window_generator.py:193-197 vs :217-231)
                    pos=al.pos_rows[lo:hi] if al.uniq is None else al.uniq[slice(*self.window_range(al.uniq, start, end))],
                    ref_gts=mats[("ref", ref_pop)][lo:hi],
                    tgt_gts=mats[("tgt", tgt_pop)][lo:hi],
                    src_gts_list=[mats[("src", s)][lo:hi] for s in src_comb],
                    out_gts=None if out_pop is None else mats[("outgroup", out_pop)][lo:hi],
                )
                yield item

    def _none_window_generator(self) -> Iterator[dict[str, Any]]:
        for ref_pop, tgt_pop, src_comb, out_pop in self.combinations():
            for start, end in self.tgt_windows[tgt_pop]:
                yield self._empty_item(ref_pop, tgt_pop, src_comb, out_pop, start, end)

    def get(self) -> Iterator[dict[str, Any]]:
        return self._window_generator() if self.has_data() else self._none_window_generator()

    def __len__(self) -> int:
        return self.total_windows
