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
from ..utils.read_data import read_data
from ..utils.windows import split_genome
from .data_generator import DataGenerator


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
    ):
        if win_len <= 0:
            raise ValueError("`win_len` must be greater than 0.")
        if win_step < 0:
            raise ValueError("`win_step` must be non-negative.")
        if num_src < 1:
            raise ValueError("`num_src` must be at least 1.")
        results = read_data(
            vcf_file=vcf_file,
            chr_name=chr_name,
            start=start,
            end=end,
            ref_ind_file=ref_ind_file,
            tgt_ind_file=tgt_ind_file,
            src_ind_file=src_ind_file,
            out_ind_file=out_ind_file,
            ploidy_config=ploidy_config,
            anc_allele_file=anc_allele_file,
        )
        self._setup(
            chr_name, win_len, win_step, ploidy_config, results["ref"], results["tgt"], results["src"], start, end, num_src,
            results["outgroup"],
        )

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

    def common_positions(self, ref_pop, tgt_pop, src_comb, out_pop=None) -> np.ndarray:
        """Positions shared by the populations of one combination.  All populations come from
        the same VCF region (and the same polarisation), so their site sets are identical; that
        is what the resident, index-range design relies on and it is checked here."""
        blocks = [self.ref_data[ref_pop], self.tgt_data[tgt_pop]] + [self.src_data[s] for s in src_comb]
        if out_pop is not None:
            blocks.append(self.out_data[out_pop])
        pos = blocks[0].POS
        for b in blocks[1:]:
            if b.POS.shape != pos.shape or not np.array_equal(b.POS, pos):
                raise NotImplementedError("populations with different site sets are not supported")
        if pos.size > 1 and not np.all(pos[1:] > pos[:-1]):
            raise NotImplementedError("positions must be strictly increasing (duplicate or unsorted POS)")
        return pos

    def device_blocks(self, eng) -> dict:
        """{(group, population): TiledPop} of every loaded population: host matrices are uploaded
        and re-tiled once per generator; blocks that already live in HBM (``from_resident``) are
        handed through."""
        from ..engine import TiledPop

        cache = self.__dict__.setdefault("_device_blocks", {})
        groups = [("ref", self.ref_data), ("tgt", self.tgt_data), ("src", self.src_data)]
        if self.out_data:
            groups.append(("outgroup", self.out_data))
        for group, data in groups:
            for pop, cd in data.items():
                if (group, pop) not in cache:
                    cache[(group, pop)] = cd.GT if isinstance(cd.GT, TiledPop) else eng.tile(cd.GT)
        return cache

    def device_positions(self, eng, pos: np.ndarray):
        """int32 device tensor of a position array (resident blocks bring their own)."""
        import torch

        dev = self.__dict__.get("_device_pos")
        if dev is not None and dev[0] is pos:
            return dev[1]
        return torch.as_tensor(np.ascontiguousarray(pos, dtype=np.int32)).to(eng.device)

    def __getstate__(self):  # device handles never travel with a pickled generator
        return {k: v for k, v in self.__dict__.items() if k not in ("_device_blocks", "_device_pos")}

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
        for ref_pop, tgt_pop, src_comb, out_pop in self.combinations():
            pos = self.common_positions(ref_pop, tgt_pop, src_comb, out_pop)
            for start, end in self.tgt_windows[tgt_pop]:
                lo, hi = self.window_range(pos, start, end)
                if hi <= lo:  # window_generator.py:199-215
                    yield self._empty_item(ref_pop, tgt_pop, src_comb, out_pop, start, end)
                    continue
                item = self._empty_item(ref_pop, tgt_pop, src_comb, out_pop, start, end)
                item.update(
                    pos=pos[lo:hi],
                    ref_gts=self.ref_data[ref_pop].GT[lo:hi],
                    tgt_gts=self.tgt_data[tgt_pop].GT[lo:hi],
                    src_gts_list=[self.src_data[s].GT[lo:hi] for s in src_comb],
                    out_gts=None if out_pop is None else self.out_data[out_pop].GT[lo:hi],
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
