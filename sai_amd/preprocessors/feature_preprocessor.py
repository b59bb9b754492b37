"""Per-window driver and output writer (mirror of
sai/preprocessors/feature_preprocessor.py:28-258).

``run`` keeps the reference's one-window plugin call (statistic classes looked up in
STAT_REGISTRY).  ``run_windows`` is the batched MI355X path the chunk driver uses: every
population block of the region is uploaded once and all windows of a population combination are
answered by ONE resident scoring pass (``score_windows`` -> ``WindowBatch``, the numeric form
that also travels between GPUs); ``items_from_batch`` then builds the item dictionaries ``run``
would have produced window by window.
"""

from __future__ import annotations

from pathlib import Path
from typing import Any, Optional

import numpy as np

from ..registries.stat_registry import STAT_REGISTRY
from .data_preprocessor import DataPreprocessor
from .window_batch import ComboBatch, WindowBatch

_HIP_STATS = ("U", "Q")
_FOURPOP = ("fd", "df", "Danc", "Dplus")
_LIST_STATS = _FOURPOP + ("DD",)  # one value per source population


def _repeated_position_semantics(res, uq_names, lo, win, uniq) -> np.ndarray:
    """What the reference does when a position occurs more than once (a VCF with split multiallelic
    records): a window's matrices keep every row (window_generator.py:217-231) but its ``pos`` is the
    UNIQUE common positions (:193-197).  N(Variants) therefore counts unique positions; U's
    ``pos[idx]`` (u_statistic.py:95) reads the k-th unique position for the k-th row -- shifted after a
    repeat, and an IndexError when k runs past the end; Q's ``pos[condition]`` (q_statistic.py:93) is
    a boolean index of the wrong length, an IndexError as soon as a window holds a repeat.

    ``res`` holds row indices in its lists (block-relative); they are replaced by those positions in
    place.  Returns the per-window unique-position counts."""
    ulo = np.searchsorted(uniq, win[:, 0], "left")
    uhi = np.searchsorted(uniq, win[:, 1], "right")
    n_uniq = (uhi - ulo).astype(np.int64)
    n_rows = res.records[0]["n_sites"].astype(np.int64)
    first_bad = None  # (window, order of the statistic, message)
    for si, name in enumerate(uq_names):
        count_field, col, flat = ("u_count", 0, res.cdd_u) if name == "U" else ("n_cdd_q", 1, res.cdd_q)
        counts = res.records[si][count_field].astype(np.int64)
        a = int(res.offsets[si, 0, col]) if counts.size else 0
        rows = flat[a : a + int(counts.sum())].astype(np.int64)
        w_of = np.repeat(np.arange(counts.size), counts)
        k = rows - lo[w_of].astype(np.int64)
        if name == "Q":
            bad = np.flatnonzero((n_rows != n_uniq) & (n_uniq > 0))
            if bad.size and (first_bad is None or (bad[0], si) < first_bad[:2]):
                w = int(bad[0])
                first_bad = (w, si, f"boolean index did not match indexed array along axis 0; size of axis is "
                                    f"{n_uniq[w]} but size of corresponding boolean axis is {n_rows[w]}")  # fmt: skip
        else:
            over = np.flatnonzero(k >= n_uniq[w_of])
            if over.size and (first_bad is None or (w_of[over[0]], si) < first_bad[:2]):
                w = int(w_of[over[0]])
                first_bad = (w, si, f"index {int(k[over[0]])} is out of bounds for axis 0 with size {int(n_uniq[w])}")
        ok = k < n_uniq[w_of]
        shifted = np.zeros(rows.size, dtype=np.int64)
        shifted[ok] = uniq[ulo[w_of[ok]] + k[ok]]
        flat[a : a + rows.size] = shifted.astype(flat.dtype)
    if first_bad is not None:
        raise IndexError(first_bad[2])
    return n_uniq.astype(np.int32)


def _rows_per_statistic(res, set_of):
    """WindowResults with one row per configured statistic from the results of the merged parameter sets
    (``set_of[i]`` = set that answers statistic i).  Rows and offsets are copied, the candidate lists are
    shared (``shared_lists``): laying them out per row costs more than everything else the host does for a
    C3 region, and only the transport between ranks needs it (``WindowResults.separate_lists``)."""
    from ..engine import WindowResults

    if list(set_of) == list(range(res.records.shape[0])):
        return res
    idx = list(set_of)
    if len(set(idx)) == 1:  # the usual configuration (U and Q of one set): read-only views, nothing copied
        shape = (len(idx), res.records.shape[1])
        return WindowResults(np.broadcast_to(res.records[idx[0]], shape), np.broadcast_to(res.offsets[idx[0]], shape + (2,)),
                             res.cdd_u, res.cdd_q, shared_lists=True)  # fmt: skip
    return WindowResults(res.records[idx], res.offsets[idx], res.cdd_u, res.cdd_q, shared_lists=True)


def _merge_window_ranges(parts):
    """WindowResults of consecutive window ranges of ONE scorer configuration -> the results over all the windows:
    records side by side, every set's lists range after range, offsets = running sums of the counts in (set, window)
    order -- the layout a single pass over all windows leaves (engine.WindowResults)."""
    from ..engine import WindowResults

    rec = np.concatenate([p.records for p in parts], axis=1)
    n_sets, n_w = rec.shape
    flat = {}
    for col, field, name in ((0, "u_count", "cdd_u"), (1, "n_cdd_q", "cdd_q")):
        pieces = []
        for s in range(n_sets):
            for p in parts:
                n = int(p.records[s][field].sum())
                a = int(p.offsets[s, 0, col]) if p.records.shape[1] else 0
                pieces.append(getattr(p, name)[a : a + n])
        flat[name] = np.concatenate(pieces) if pieces else getattr(parts[0], name)[:0]
    off = np.zeros((n_sets, n_w, 2), dtype=np.int64)
    for col, field in ((0, "u_count"), (1, "n_cdd_q")):
        counts = rec[field].reshape(-1).astype(np.int64)
        off[:, :, col] = (np.cumsum(counts) - counts).reshape(n_sets, n_w)
    return WindowResults(rec, off, flat["cdd_u"], flat["cdd_q"])


def _file_order_semantics(res, uq_names, lo, hi, pos_sorted, file_order) -> None:
    """What the reference reports when the positions of a region do not ascend (an unsorted VCF): its
    window matrices keep the FILE order of the rows (``GT.compress`` of a mask, window_generator.py:217-231)
    while ``pos`` is intersect1d's SORTED array (:193-197), so U's ``pos[idx]`` and Q's ``pos[condition]``
    name, for the k-th row of the window in file order, the k-th smallest position.  The counts and Q do
    not depend on the order.  ``res`` holds block-relative indices of the position-sorted rows in its
    lists; they are replaced in place by those positions, each list ascending as the reference's."""
    for si, name in enumerate(uq_names):
        count_field, col, flat = ("u_count", 0, res.cdd_u) if name == "U" else ("n_cdd_q", 1, res.cdd_q)
        counts = res.records[si][count_field].astype(np.int64)
        at = int(res.offsets[si, 0, col]) if counts.size else 0
        first = np.cumsum(counts) - counts
        for w in np.flatnonzero(counts):
            a, b = int(lo[w]), int(hi[w])
            rows = flat[at + int(first[w]) : at + int(first[w] + counts[w])]
            in_file = file_order[a:b]  # file index of every sorted row of the window
            rank = np.empty(b - a, dtype=np.int64)
            rank[np.argsort(in_file, kind="stable")] = np.arange(b - a)  # k of each sorted row
            rows[:] = np.sort(pos_sorted[a + rank[rows.astype(np.int64) - a]]).astype(rows.dtype)


class FeaturePreprocessor(DataPreprocessor):
    def __init__(self, output_file: str, stat_config, anc_allele_available: bool = False):
        self.output_file = output_file
        self.anc_allele_available = anc_allele_available
        self.stat_config = stat_config

    # -- helpers ---------------------------------------------------------------------------

    def _active_stats(self) -> list[str]:
        """Statistic names in config order: U/Q always, the others when set to True
        (feature_preprocessor.py:146-151)."""
        return [
            name
            for name, value in self.stat_config.root.items()
            if name in _HIP_STATS or (name in _LIST_STATS and value is True)
        ]

    @staticmethod
    def _new_item(chr_name, start, end, ref_pop, tgt_pop, src_pop_list, out_pop, nsnps) -> dict[str, Any]:
        # feature_preprocessor.py:113-129
        return {
            "chr_name": chr_name,
            "start": start,
            "end": end,
            "ref_pop": ref_pop,
            "tgt_pop": tgt_pop,
            "src_pop_list": src_pop_list,
            "out_pop": "NA" if out_pop is None else out_pop,
            "nsnps": nsnps,
            "cdd_pos": {},
        }

    def _fill_missing(self, item: dict, names) -> None:
        # feature_preprocessor.py:131-144
        n_src = len(item["src_pop_list"])
        for name in names:
            if name in _HIP_STATS:
                item[name] = np.nan
                item["cdd_pos"][name] = np.array([])
            else:
                item[name] = [np.nan for _ in range(n_src)] if n_src > 1 else np.nan

    def _stat_kwargs(self, name: str, ref_pop: str, tgt_pop: str) -> dict:
        # feature_preprocessor.py:164-185: thresholds per (ref_pop, tgt_pop), sources by position
        prm = self.stat_config.get_parameters(name)
        kw = dict(
            w=prm["ref"][ref_pop],
            y_list=list(prm["src"].values()),
            anc_allele_available=self.anc_allele_available,
        )
        if name == "U":
            kw["x"] = prm["tgt"][tgt_pop]
        else:
            kw["quantile"] = prm["tgt"][tgt_pop]
        return kw

    # -- one window through the plugin API -------------------------------------------------

    def run(
        self,
        chr_name: str,
        ref_pop: str,
        tgt_pop: str,
        src_pop_list: list[str],
        out_pop: Optional[str],
        start: int,
        end: int,
        pos: np.ndarray,
        ref_gts: np.ndarray,
        tgt_gts: np.ndarray,
        src_gts_list: list[np.ndarray],
        out_gts: Optional[np.ndarray],
        ploidy_config,
    ) -> list[dict[str, Any]]:
        names = self._active_stats()
        item = self._new_item(chr_name, start, end, ref_pop, tgt_pop, src_pop_list, out_pop, len(pos))
        if ref_gts is None or tgt_gts is None or src_gts_list is None or ploidy_config is None:
            self._fill_missing(item, names)
            return [item]
        from ..engine import Engine

        # one upload of the window's matrices serves every configured statistic, and U and Q of one parameter set
        # are one device call (a record carries both): the scope is told which thresholds go together
        hints = {}
        try:
            pl = (ploidy_config.get_ploidy("ref", ref_pop), ploidy_config.get_ploidy("tgt", tgt_pop), *ploidy_config.get_ploidy("src"))
            for name in names:
                if name in _HIP_STATS:
                    kw = self._stat_kwargs(name, ref_pop, tgt_pop)
                    key = (tuple(int(p) for p in pl[: 2 + len(src_gts_list)]), float(kw["w"]),
                           tuple((op, float(y)) for op, y in kw["y_list"]), bool(kw["anc_allele_available"]))  # fmt: skip
                    hints.setdefault(key, {})["x" if name == "U" else "quantile"] = float(kw["x"] if name == "U" else kw["quantile"])
        except (KeyError, TypeError, ValueError):  # whatever is wrong with the configuration is the statistics' to report
            hints = {}
        with Engine.get().upload_scope(hints):
            self._run_statistics(item, names, ref_pop, tgt_pop, pos, ref_gts, tgt_gts, src_gts_list, out_gts, out_pop,
                                 ploidy_config)  # fmt: skip
        return [item]

    def _run_statistics(self, item, names, ref_pop, tgt_pop, pos, ref_gts, tgt_gts, src_gts_list, out_gts, out_pop,
                        ploidy_config) -> None:  # fmt: skip
        for name in names:
            stat = STAT_REGISTRY.get(name)(
                ref_gts=ref_gts,
                tgt_gts=tgt_gts,
                src_gts_list=src_gts_list,
                out_gts=out_gts,
                ref_ploidy=ploidy_config.get_ploidy("ref", ref_pop),
                tgt_ploidy=ploidy_config.get_ploidy("tgt", tgt_pop),
                src_ploidy_list=ploidy_config.get_ploidy("src"),
                out_ploidy=ploidy_config.get_ploidy("outgroup", out_pop),
            )
            if name in _HIP_STATS:
                res = stat.compute(pos=pos, **self._stat_kwargs(name, ref_pop, tgt_pop))
                item["cdd_pos"][name] = res["cdd_pos"]
            else:
                res = stat.compute()
            item[name] = res["value"]

    # -- all windows of a resident region, batched on the GPU -------------------------------

    def run_windows(self, wg) -> list[dict[str, Any]]:
        """Items for every (population combination, window) of a WindowGenerator, in its order."""
        return self.items_from_batch(self.score_windows(wg))

    def score_and_write(self, wg) -> None:
        """``write_batches([score_windows(wg)])`` for the ONE chunk of a run, with the two overlapped: a large region
        whose combination is served by the fused pass alone is scored in ``PARTS`` contiguous window ranges -- each
        over its own tile range of the resident blocks, the passes queued back to back -- and the rows of a part
        are formatted and written while the GPU streams the following parts (the reference writes as its workers
        deliver, sai.py:146-151, feature_preprocessor.py:193-258).  Same bytes as the two calls."""
        files = self._open_outputs()
        try:
            self.score_windows(wg, sink=lambda cb: self._write_combo(files, wg.chr_name, cb))
        finally:
            for fh in files:
                fh.close()

    PARTS = 3  # window ranges a large region is scored in when its rows are written as they arrive (3: 3.58 ms for C3 on one box, 4: 3.70-3.97, 6: 4.18, one: 4.08-4.12 -- profiles/r05_score_parts.txt)
    PART_MIN_WINDOWS = 2048
    PART_FRACTIONS = None  # where the ranges end, as fractions of the windows (PARTS - 1 of them); None: equal ranges but a last one of 0.6 of their size

    def score_windows(self, wg, sink=None) -> WindowBatch:
        """The GPU part of ``run_windows``: every population block of the region is uploaded once
        and U / Q are answered by a ``ResidentScorer`` -- the fused site pass (genotypes streamed
        once, per-site decision in the same launch, tgt_freq only at candidate sites) followed by
        the windows stage -- i.e. by the very path bench.py times.  When several population
        combinations share blocks, or the ABBA-BABA family needs the per-population counts anyway,
        each block is reduced once (site_counts) and the combinations start from those counts.

        ``sink(ComboBatch)``: called with every combination's results as soon as they are on the host -- a large
        one in several parts, each a ComboBatch over a contiguous range of its windows (``score_and_write``); the
        returned batch then lists the parts as they were handed over."""
        names = self._active_stats()
        uq_names = [n for n in names if n in _HIP_STATS]
        four_names = [n for n in names if n in _FOURPOP]
        want_dd = "DD" in names
        combos = list(wg.combinations())
        batch = WindowBatch(wg.chr_name, [])
        if not wg.has_data():
            for ref_pop, tgt_pop, src_comb, out_pop in combos:
                win = np.asarray(wg.tgt_windows[tgt_pop], dtype=np.int64).reshape(-1, 2)
                batch.combos.append(ComboBatch(ref_pop, tgt_pop, tuple(src_comb), out_pop, win, np.zeros(len(win), np.int32)))
                if sink is not None:
                    sink(batch.combos[-1])
            return batch

        import torch

        from .. import _ffi
        from ..engine import Engine
        from ..resident import ResidentBlock, ResidentScorer
        from ..stats.stat_utils import _check_ploidy, validate_thresholds

        eng = Engine.get()
        pc = wg.ploidy_config
        shared_tiled = wg.device_blocks(eng)  # {(group, population): TiledPop}, uploaded (or already resident) once
        shared = len(combos) > 1 or bool(four_names) or len(uq_names) > _ffi.SAI_FUSED_SETS
        shared_counts = {}
        tiled, counts_rows = shared_tiled, shared_counts

        def counts_of(keys):
            """int32 [len(keys)][n_sites][2]; every block is reduced at most once."""
            todo = [k for k in dict.fromkeys(keys) if k not in counts_rows]
            max_pops = 2 + _ffi.SAI_FUSED_SRC
            for i in range(0, len(todo), max_pops):
                part = todo[i : i + max_pops]
                c = eng.site_counts([tiled[k] for k in part])
                for j, k in enumerate(part):
                    counts_rows[k] = c[j]
            return torch.stack([counts_rows[k] for k in keys])

        pos_dev_cache = {}
        win_arrays = wg.__dict__.setdefault("_win_arrays", {})  # the grid as int64 [n][2], once per generator (0.7 ms for 10^4 tuples)
        scorers = wg.__dict__.setdefault("_scorers", {})
        group_data = {"ref": wg.ref_data, "tgt": wg.tgt_data, "src": wg.src_data, "outgroup": wg.out_data}
        for ref_pop, tgt_pop, src_comb, out_pop in combos:
            al = wg.aligned(ref_pop, tgt_pop, src_comb, out_pop)
            pos = al.pos_rows
            if al.rows is not None:
                # populations with different site sets (window_generator.py:193-231): this combination
                # works on its own row-gathered copies of the blocks
                tiled = dict(zip(al.keys, eng.tile_many([group_data[g][p].GT[al.rows[(g, p)]] for g, p in al.keys])))
                counts_rows = {}
            else:
                tiled, counts_rows = shared_tiled, shared_counts
            win = win_arrays.get(tgt_pop)
            if win is None:  # one conversion per target population, not per combination
                win = win_arrays[tgt_pop] = np.asarray(wg.tgt_windows[tgt_pop], dtype=np.int64).reshape(-1, 2)
            windows = win
            src_ploidies = pc.get_ploidy("src")
            ploidy = [pc.get_ploidy("ref", ref_pop), pc.get_ploidy("tgt", tgt_pop)] + list(src_ploidies)
            n_eff = min(len(src_comb), len(src_ploidies))
            # U and Q that share w, the source conditions and the polarity mode are ONE parameter set (a record
            # carries the U count and Q): the usual configuration then costs one evaluation and one windows stage
            sets, set_of, merged = [], [], {}
            for name in uq_names:
                kw = self._stat_kwargs(name, ref_pop, tgt_pop)
                validate_thresholds(kw["w"], kw["y_list"], len(src_comb))
                key = (float(kw["w"]), tuple((op, float(y)) for op, y in kw["y_list"]), bool(kw["anc_allele_available"]))
                field, value = ("x", kw["x"]) if name == "U" else ("quantile", kw["quantile"])
                at = next((i for i in merged.get(key, ()) if field not in sets[i][1]), None)
                if at is None:
                    at = len(sets)
                    sets.append((kw, {}))
                    merged.setdefault(key, []).append(at)
                sets[at][1][field] = value
                set_of.append(at)
            sets = [_ffi.make_params(kw["w"], got.get("x", 0.0), got.get("quantile", 0.5), kw["y_list"],
                                     kw["anc_allele_available"], n_src=n_eff) for kw, got in sets]  # fmt: skip
            for p in ploidy[: 2 + len(src_comb)]:
                _check_ploidy(p)
            n_sites = int(pos.size)
            cb = ComboBatch(ref_pop, tgt_pop, tuple(src_comb), out_pop, win, np.zeros(len(win), np.int32), list(uq_names),
                            pos_dtype=np.dtype(pos.dtype).name)  # fmt: skip
            if not (names and len(win) and n_sites):
                batch.combos.append(cb)
                if sink is not None:
                    sink(cb)
                continue
            if (sink is not None and uq_names and not shared and not four_names and not want_dd and al.segments is None
                    and al.uniq is None and al.file_order is None and len(win) >= self.PART_MIN_WINDOWS
                    and n_eff <= _ffi.SAI_FUSED_SRC):  # fmt: skip
                self._score_in_parts(eng, wg, cb, al, tiled, ploidy, n_eff, sets, set_of, sink, batch)
                continue
            batch.combos.append(cb)
            pid = id(pos)
            if pid not in pos_dev_cache:
                pos_dev_cache[pid] = wg.device_positions(eng, pos)
            pos_dev = pos_dev_cache[pid]
            uq_keys = [("ref", ref_pop), ("tgt", tgt_pop)] + [("src", s) for s in src_comb[:n_eff]]
            lo = hi = None
            segs = al.segments  # one piece per window: populations with different unsorted position arrays
            # DD's per-site terms ride along a pass over ref and tgt that is needed anyway (Engine.site_pass_dd):
            # the scorer's fused pass, or the counts pass of the combination's blocks -- the genotypes are then
            # read once, not once more per two source individuals (dd_statistic.py:60-77)
            dd_keys = [("ref", ref_pop), ("tgt", tgt_pop)] + [("src", s) for s in src_comb]
            dd_terms = None
            dd_along = want_dd and eng.dd_rides_along([tiled[k] for k in dd_keys], 2, len(src_comb))
            if dd_along:
                rows = sum(tiled[k].n_ind for k in dd_keys[2:])
                dd_terms = torch.empty((2, rows, n_sites), dtype=torch.int32, device=eng.device)
            dd_in_scorer = dd_along and bool(uq_names) and not shared and n_eff == len(src_comb)
            if dd_along and not dd_in_scorer:
                keys = list(dd_keys)
                if four_names and out_pop is not None:
                    keys.append(("outgroup", out_pop))
                c = torch.empty((len(keys), n_sites, 2), dtype=torch.int32, device=eng.device) if shared else None
                eng.site_pass_dd([tiled[k] for k in keys], None, [], 2, len(src_comb), counts=c, absdiff=dd_terms)
                if shared:
                    for j, k in enumerate(keys):
                        counts_rows.setdefault(k, c[j])
            if uq_names:
                block = ResidentBlock([tiled[k] for k in uq_keys], ploidy[: 2 + n_eff], pos_dev,
                                      segments=None if segs is None else [tuple(map(int, sg)) for sg in segs],
                                      extra={"output_arena": wg.__dict__.get("_output_arena") if tiled is shared_tiled else None})  # fmt: skip
                # one scorer per (window grid, block length, number of sets) serves every combination of the
                # region -- and the next call on the same generator: only its launch sequences are re-recorded
                key = (tgt_pop, n_sites, len(sets)) if segs is None else (tgt_pop, n_sites, len(sets), tuple(al.keys))
                as_indices = al.uniq is not None or al.file_order is not None
                dd_arg = (2, len(src_comb), dd_terms) if dd_in_scorer else None
                scorer = scorers.get(key)
                if scorer is None:
                    scorer = scorers[key] = ResidentScorer(eng, block, windows, sets, cap_u=1 << 16, cap_q=1 << 16,
                                                           counts_in=counts_of(uq_keys) if shared else None,
                                                           lists_as_indices=as_indices, fetch_lists=1 << 16,
                                                           window_segment=None if segs is None else np.arange(len(segs)),
                                                           dd_out=dd_arg)  # fmt: skip
                else:
                    scorer.rebind(block, sets, counts_of(uq_keys) if shared else None, lists_as_indices=as_indices,
                                  dd_out=dd_arg)  # fmt: skip
                scorer.step()
                cb.uq = _rows_per_statistic(scorer.results(grow=True), set_of)
                lo, hi = scorer.lo, scorer.hi
                cb.nsnps = cb.uq.records[0]["n_sites"].astype(np.int32)
                if al.uniq is not None:
                    cb.nsnps = _repeated_position_semantics(cb.uq, uq_names, lo.cpu().numpy(), win, al.uniq)
                elif al.file_order is not None:
                    _file_order_semantics(cb.uq, uq_names, lo.cpu().numpy(), hi.cpu().numpy(), pos, al.file_order)
            elif segs is not None:  # the windows' site ranges ARE the pieces
                bounds = np.asarray(segs, dtype=np.int32).reshape(-1, 2)
                lo, hi = (torch.from_numpy(np.ascontiguousarray(bounds[:, k])).to(eng.device) for k in (0, 1))
                cb.nsnps = (bounds[:, 1] - bounds[:, 0]).astype(np.int32)
            else:
                lo, hi = eng.window_bounds(pos_dev, win[:, 0], win[:, 1])
                cb.nsnps = (hi - lo).cpu().numpy().astype(np.int32)
                if al.uniq is not None:
                    cb.nsnps = (np.searchsorted(al.uniq, win[:, 1], "right") - np.searchsorted(al.uniq, win[:, 0], "left")).astype(np.int32)
            if four_names:  # every source of the combination, with its own ploidy (fd_statistic.py:63-74)
                if len(src_ploidies) < len(src_comb):
                    raise IndexError("list index out of range")
                keys = [("ref", ref_pop), ("tgt", tgt_pop)] + [("src", s) for s in src_comb]
                pl4 = [ploidy[0], ploidy[1]] + list(src_ploidies[: len(src_comb)])
                if out_pop is not None:
                    keys.append(("outgroup", out_pop))
                    pl4.append(pc.get_ploidy("outgroup", out_pop))
                for p in pl4:
                    _check_ploidy(p)
                cb.four = eng.fourpop_windows(counts_of(keys), pl4, len(src_comb), out_pop is not None, lo, hi).cpu().numpy()
            if dd_terms is not None:  # the terms came with the pass: only the window sums are left
                ref_t, tgt_t = tiled[("ref", ref_pop)], tiled[("tgt", tgt_pop)]
                cols, row = [], 0
                for s in src_comb:
                    n = tiled[("src", s)].n_ind
                    cols.append(eng.window_dd(dd_terms[0, row : row + n], ref_t.n_ind, dd_terms[1, row : row + n], tgt_t.n_ind,
                                              lo, hi).cpu().numpy())  # fmt: skip
                    row += n
                cb.dd = np.stack(cols, axis=1)
            elif want_dd:  # more source individuals than ride along: two streaming passes per pair of them
                ref_t, tgt_t = tiled[("ref", ref_pop)], tiled[("tgt", tgt_pop)]
                cb.dd = np.stack(
                    [
                        eng.window_dd(eng.site_absdiff(ref_t, tiled[("src", s)]), ref_t.n_ind,
                                      eng.site_absdiff(tgt_t, tiled[("src", s)]), tgt_t.n_ind, lo, hi).cpu().numpy()
                        for s in src_comb
                    ],
                    axis=1,
                )  # fmt: skip
            if sink is not None:
                sink(cb)
        return batch

    def _score_in_parts(self, eng, wg, cb, al, tiled, ploidy, n_eff, sets, set_of, sink, batch) -> None:
        """One combination of a large region as ``PARTS`` contiguous window ranges.  A part needs the sites from
        its first window's start to its last window's end: the tiles that hold them are a contiguous slice of
        every population's tiled block (a tile is ``n_ind * 64`` bytes), so each part is a scorer of its own over
        views -- nothing is copied, the halo between neighbouring parts (a window length of sites) is read twice.
        All passes are enqueued first; the parts' results are then taken in order, each as soon as ITS windows
        stage has delivered, and handed to ``sink`` while the later passes still stream."""
        from .. import _ffi
        from ..engine import TiledPop
        from ..resident import ResidentBlock, ResidentScorer
        from ..utils.windows import split_index_ranges

        pos, win = al.pos_rows, cb.windows
        n_sites, tile = int(pos.size), _ffi.SAI_TILE_SITES
        keys = [("ref", cb.ref_pop), ("tgt", cb.tgt_pop)] + [("src", s) for s in cb.src_comb[:n_eff]]
        # the parts of a (generator, combination, parameter sets) are laid out once: a later call on the same
        # generator -- the same region scored again -- finds its scorers bound and only enqueues their passes
        plans = wg.__dict__.setdefault("_part_plans", {})
        plan_key = (cb.tgt_pop, cb.ref_pop, tuple(cb.src_comb), self.PARTS, self.PART_FRACTIONS)
        signature = ResidentScorer._binding_signature(ResidentBlock([tiled[k] for k in keys], ploidy[: 2 + n_eff], wg.device_positions(eng, pos)),
                                                      sets, None, False)  # fmt: skip
        plan = plans.get(plan_key)
        if plan is None or plan[0] != signature:
            pos_dev = wg.device_positions(eng, pos)
            scorers = wg.__dict__.setdefault("_scorers", {})
            # the last part is the smallest: its rows are written after the GPU has finished
            n_w = len(win)
            fractions = self.PART_FRACTIONS or np.cumsum([1.0 / (self.PARTS - 0.4)] * (self.PARTS - 1))
            cuts = [0] + [int(round(n_w * f)) for f in fractions] + [n_w]
            ranges = [(a, b) for a, b in zip(cuts, cuts[1:]) if b > a] if self.PARTS > 1 else split_index_ranges(n_w, 1)
            built = []
            for k, (w0, w1) in enumerate(ranges):
                # needles in the positions' own dtype: numpy would otherwise convert the whole array for every search
                info = np.iinfo(pos.dtype)
                first, last = (pos.dtype.type(min(max(int(v), info.min), info.max)) for v in (win[w0, 0], win[w1 - 1, 1]))
                s_lo, s_hi = int(np.searchsorted(pos, first, "left")), int(np.searchsorted(pos, last, "right"))
                t0, t1 = s_lo // tile, max(-(-s_hi // tile), s_lo // tile + 1)
                a, b = t0 * tile, min(t1 * tile, n_sites)
                pops = [TiledPop(tiled[key].tiles[t0 * tiled[key].n_ind * tile : t1 * tiled[key].n_ind * tile], b - a, tiled[key].n_ind)
                        for key in keys]  # fmt: skip
                # (the arena belongs to the settled blocks of the generator, not to re-tiled row selections)
                block = ResidentBlock(pops, ploidy[: 2 + n_eff], pos_dev[a:b],
                                      extra={"output_arena": wg.__dict__.get("_output_arena") if tiled is wg.__dict__.get("_device_blocks") else None})  # fmt: skip
                key = (cb.tgt_pop, n_sites, len(sets), "part", k, self.PARTS, self.PART_FRACTIONS)
                scorer = scorers.get(key)
                if scorer is None or scorer.block.n_sites != block.n_sites or scorer.n_windows != w1 - w0:
                    scorer = scorers[key] = ResidentScorer(eng, block, win[w0:w1], sets, cap_u=1 << 16, cap_q=1 << 16, fetch_lists=1 << 16)
                else:
                    scorer.rebind(block, sets)
                built.append((scorer, w0, w1))
            plan = plans[plan_key] = (signature, built)
        running = plan[1]
        for scorer, _, _ in running:
            scorer.step()
        # Rows are written part by part only when that is cheap: with long candidate lists (a loose sweep: tens of
        # entries per window) the writer's calls cost more apiece than the passes they would hide behind (C5's first
        # set: 4.0 ms in three calls, 1.6 in one -- profiles/r05_score_parts.txt), so the parts are then put together
        # and written once.  The first part's lists decide.
        held, streaming = [], None
        for scorer, w0, w1 in running:
            raw = scorer.results(grow=True)
            if streaming is None:
                streaming = (raw.cdd_u.size + raw.cdd_q.size) <= self.PART_MAX_LIST_ENTRIES_PER_WINDOW * (w1 - w0)
            if not streaming:
                held.append(raw)
                continue
            part = ComboBatch(cb.ref_pop, cb.tgt_pop, cb.src_comb, cb.out_pop, win[w0:w1], None, list(cb.uq_names), pos_dtype=cb.pos_dtype)
            part.uq = _rows_per_statistic(raw, set_of)
            part.nsnps = part.uq.records[0]["n_sites"].astype(np.int32)
            batch.combos.append(part)
            sink(part)
        if held:
            cb.uq = _rows_per_statistic(_merge_window_ranges(held), set_of)
            cb.nsnps = cb.uq.records[0]["n_sites"].astype(np.int32)
            batch.combos.append(cb)
            sink(cb)

    PART_MAX_LIST_ENTRIES_PER_WINDOW = 8

    def items_from_batch(self, batch: WindowBatch, combos=None) -> list[dict[str, Any]]:
        """The reference's item dictionaries (feature_preprocessor.py:113-191) of a batch, in
        (combination, window) order; ``combos`` restricts it to some ComboBatch objects."""
        names = self._active_stats()
        four_names = [n for n in names if n in _FOURPOP]
        items: list[dict[str, Any]] = []
        for cb in batch.combos if combos is None else combos:
            pos_dtype = np.dtype(cb.pos_dtype)
            src_comb = tuple(cb.src_comb)
            starts, ends, nsnps_all = cb.windows[:, 0].tolist(), cb.windows[:, 1].tolist(), cb.nsnps.tolist()
            res = cb.uq
            if res is not None:
                u_count, n_cond = res.records["u_count"].tolist(), res.records["n_cond"].tolist()
                n_cdd_q, q_vals, off = res.records["n_cdd_q"].tolist(), res.records["q"], res.offsets.tolist()
                cdd_u, cdd_q = res.cdd_u.astype(pos_dtype, copy=False), res.cdd_q.astype(pos_dtype, copy=False)
            for wi, (start, end, nsnps) in enumerate(zip(starts, ends, nsnps_all)):
                item = self._new_item(batch.chr_name, start, end, cb.ref_pop, cb.tgt_pop, src_comb, cb.out_pop, nsnps)
                if nsnps == 0:  # window without sites: the reference's None-matrix branch
                    self._fill_missing(item, names)
                    items.append(item)
                    continue
                for name in four_names:  # one value per source, Python floats like the reference
                    item[name] = [float(v) for v in cb.four[wi, :, _FOURPOP.index(name)]]
                if cb.dd is not None:  # np.float64 values, as np.mean returns them (dd_statistic.py:74)
                    item["DD"] = [np.float64(v) for v in cb.dd[wi]]
                for si, name in enumerate(cb.uq_names):
                    if name == "U":
                        o = off[si][wi][0]
                        item["cdd_pos"][name] = cdd_u[o : o + u_count[si][wi]].copy()
                        item[name] = u_count[si][wi]
                    elif n_cond[si][wi] == 0:
                        item["cdd_pos"][name] = np.array([])
                        item[name] = np.nan
                    else:
                        o = off[si][wi][1]
                        item["cdd_pos"][name] = cdd_q[o : o + n_cdd_q[si][wi]].copy()
                        item[name] = q_vals[si, wi]
                items.append(item)
        return items

    def items_from_batches(self, batches) -> list[dict[str, Any]]:
        """Items of several chunks of ONE chromosome region list, in the order a single-chunk run
        emits them: combination-major, and inside a combination the chunks' windows in chunk order
        (sai.py:146-151 with ``num_chunks=1``) -- so a sharded run writes the same bytes."""
        batches = list(batches)
        if not batches:
            return []
        n_combos = len(batches[0].combos)
        if any(len(b.combos) != n_combos for b in batches):
            raise ValueError("chunks disagree about the population combinations")
        items: list[dict[str, Any]] = []
        for k in range(n_combos):
            for b in batches:
                items.extend(self.items_from_batch(b, [b.combos[k]]))
        return items

    # -- output ----------------------------------------------------------------------------

    def _open_outputs(self) -> list:
        """The TSV and the .U.log / .Q.log files, opened for appending.  Unbuffered: the library writes to the
        descriptors itself (one fan-out formats a piece of every file, writev() in window order), nothing of the
        text passes through Python."""
        files = [open(self.output_file, "ab", buffering=0)]
        try:
            for key in _HIP_STATS:
                if key in self.stat_config.root:
                    files.append(open(Path(self.output_file).with_suffix(f".{key}.log"), "ab", buffering=0))
        except BaseException:
            for fh in files:
                fh.close()
            raise
        return files

    def _write_combo(self, files, chr_name, cb) -> None:
        """The rows of one ComboBatch (a combination, or a window range of one) behind what the files hold."""
        import ctypes as C

        from .. import _ffi

        lib = _ffi.load_host()
        names = self._active_stats()
        log_keys = [key for key in _HIP_STATS if key in self.stat_config.root]
        keep = []  # arrays the column descriptors point into

        def column(arr, kind):
            keep.append(arr)
            return _ffi.SaiTextColumn(arr.ctypes.data, arr.strides[0] if arr.ndim else 0, kind, 0)

        n_w = int(cb.windows.shape[0])
        win = np.ascontiguousarray(cb.windows, dtype=np.int64)
        nsnps = np.ascontiguousarray(cb.nsnps, dtype=np.int32)
        n_src = len(cb.src_comb)
        zeros = np.zeros(max(n_w, 1), dtype=np.float64)
        cols = []
        for name in names:
            if name in _HIP_STATS:
                if cb.uq is None:
                    cols.append(column(zeros, 1))
                    continue
                si = cb.uq_names.index(name)
                rec = cb.uq.records[si]
                cols.append(column(rec["u_count"], 0) if name == "U" else column(rec["q"], 1))
            elif name == "DD":
                for s_i in range(max(n_src, 1)):
                    cols.append(column(zeros if cb.dd is None else cb.dd[:, s_i], 1))
            else:
                for s_i in range(max(n_src, 1)):
                    cols.append(column(zeros if cb.four is None else cb.four[:, s_i, _FOURPOP.index(name)], 1))
        arr = (_ffi.SaiTextColumn * max(len(cols), 1))(*cols)
        pops = f"{cb.ref_pop}\t{cb.tgt_pop}\t{','.join(cb.src_comb)}\t{'NA' if cb.out_pop is None else cb.out_pop}"
        logs = (_ffi.SaiLogRows * max(len(log_keys), 1))()
        for i, key in enumerate(log_keys):
            if cb.uq is None or key not in cb.uq_names:
                counts, offs, lists = np.zeros(max(n_w, 1), dtype=np.int32), np.zeros(max(n_w, 1), dtype=np.int64), None
            else:
                si = cb.uq_names.index(key)
                counts = cb.uq.records[si]["u_count" if key == "U" else "n_cdd_q"]
                offs = np.ascontiguousarray(cb.uq.offsets[si, :, 0 if key == "U" else 1])
                lists = np.ascontiguousarray(cb.uq.cdd_u if key == "U" else cb.uq.cdd_q)
            keep.extend([counts, offs, lists])
            logs[i] = _ffi.SaiLogRows(counts.ctypes.data, counts.strides[0], offs.ctypes.data, 1,
                                      None if lists is None or lists.size == 0 else lists.ctypes.data,
                                      4 if lists is None else lists.dtype.itemsize, files[1 + i].fileno())  # fmt: skip
        _ffi.check(lib.sai_write_window_rows(str(chr_name).encode(), pops.encode(), n_w, C.c_void_p(win.ctypes.data),
                                             C.c_void_p(nsnps.ctypes.data), len(cols), arr, files[0].fileno(),
                                             len(log_keys), logs, None), lib)  # fmt: skip

    def write_batches(self, batches) -> None:
        """``process_items(items_from_batches(batches))`` without the items: the TSV and log rows are
        formatted natively (libsaihip ``sai_write_window_rows``) straight from the records, the CSR candidate
        lists and the f64 blocks -- same bytes, same order (combination-major, the chunks' windows in chunk
        order), about a fifth of the host time."""
        batches = list(batches)
        if not batches:
            return
        n_combos = len(batches[0].combos)
        if any(len(b.combos) != n_combos for b in batches):
            raise ValueError("chunks disagree about the population combinations")
        files = self._open_outputs()
        try:
            for k in range(n_combos):
                for b in batches:
                    self._write_combo(files, b.chr_name, b.combos[k])
        finally:
            for f in files:
                f.close()

    def process_items(self, items: list[dict[str, Any]]) -> None:
        """Append TSV rows and ``.U.log`` / ``.Q.log`` rows (feature_preprocessor.py:193-258):
        values through ``str()``; candidate lists as comma-joined ``chrom:pos`` or ``NA``."""
        names = self._active_stats()
        with open(self.output_file, "a") as f:
            for item in items:
                vals = []
                for name in names:  # feature_preprocessor.py:217-228
                    v = item.get(name)
                    if isinstance(v, list) and len(v) == len(item["src_pop_list"]):
                        vals.extend("" if x is None else str(x) for x in v)
                    else:
                        if isinstance(v, list):
                            v = v[0] if len(v) > 0 else ""
                        vals.append("" if v is None else str(v))
                f.write(
                    f"{item['chr_name']}\t{item['start']}\t{item['end']}\t{item['ref_pop']}\t{item['tgt_pop']}\t"
                    f"{','.join(item['src_pop_list'])}\t{item['out_pop']}\t{item['nsnps']}\t" + "\t".join(vals) + "\n"
                )
        for key in _HIP_STATS:
            if key not in self.stat_config.root:
                continue
            log_file = Path(self.output_file).with_suffix(f".{key}.log")
            with open(log_file, "a") as f:
                for item in items:
                    cdd = item["cdd_pos"][key]
                    txt = "NA" if cdd.size == 0 else ",".join(f"{item['chr_name']}:{p}" for p in cdd)
                    f.write(f"{item['chr_name']}\t{item['start']}\t{item['end']}\t{txt}\n")
