"""Per-window driver and output writer (mirror of
sai/preprocessors/feature_preprocessor.py:28-258).

``run`` keeps the reference's one-window plugin call (statistic classes looked up in
STAT_REGISTRY).  ``run_windows`` is the batched MI355X path the chunk driver uses: every
population block of the region is uploaded and reduced once (site_counts), and all windows of
all population combinations are answered by a handful of launches; the item dictionaries it
returns are the ones ``run`` would have produced window by window.
"""

from __future__ import annotations

from pathlib import Path
from typing import Any, Optional

import numpy as np

from ..registries.stat_registry import STAT_REGISTRY
from .data_preprocessor import DataPreprocessor

_HIP_STATS = ("U", "Q")
_FOURPOP = ("fd", "df", "Danc", "Dplus")
_LIST_STATS = _FOURPOP + ("DD",)  # one value per source population


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
        return [item]

    # -- all windows of a resident region, batched on the GPU -------------------------------

    def run_windows(self, wg) -> list[dict[str, Any]]:
        """Items for every (population combination, window) of a WindowGenerator, in its order."""
        names = self._active_stats()
        items: list[dict[str, Any]] = []
        if not wg.has_data():
            for w in wg.get():
                item = self._new_item(w["chr_name"], w["start"], w["end"], w["ref_pop"], w["tgt_pop"],
                                      w["src_pop_list"], w["out_pop"], 0)  # fmt: skip
                self._fill_missing(item, names)
                items.append(item)
            return items

        import torch

        from .. import _ffi
        from ..engine import Engine
        from ..stats.stat_utils import _check_ploidy, validate_thresholds

        eng = Engine.get()
        pc = wg.ploidy_config
        # upload + reduce every population block once
        blocks = {}
        groups = [("ref", wg.ref_data), ("tgt", wg.tgt_data), ("src", wg.src_data)]
        if wg.out_data:
            groups.append(("outgroup", wg.out_data))
        for group, data in groups:
            for pop, cd in data.items():
                blocks[(group, pop)] = eng.tile(cd.GT)
        keys = list(blocks)
        counts_rows = {}
        max_pops = 2 + _ffi.SAI_MAX_SRC
        for i in range(0, len(keys), max_pops):
            part = keys[i : i + max_pops]
            counts = eng.site_counts([blocks[k] for k in part])
            for j, k in enumerate(part):
                counts_rows[k] = counts[j]
        tiled = blocks if "DD" in names else None  # DD streams the genotype blocks again
        del blocks

        pos_dev_cache = {}
        for ref_pop, tgt_pop, src_comb, out_pop in wg.combinations():
            pos = wg.common_positions(ref_pop, tgt_pop, src_comb, out_pop)
            windows = wg.tgt_windows[tgt_pop]
            src_ploidies = pc.get_ploidy("src")
            ploidy = [pc.get_ploidy("ref", ref_pop), pc.get_ploidy("tgt", tgt_pop)] + list(src_ploidies)
            n_eff = min(len(src_comb), len(src_ploidies))
            uq_names = [n for n in names if n in _HIP_STATS]
            four_names = [n for n in names if n in _FOURPOP]
            want_dd = "DD" in names
            sets, kwargs = [], {}
            for name in uq_names:
                kw = self._stat_kwargs(name, ref_pop, tgt_pop)
                validate_thresholds(kw["w"], kw["y_list"], len(src_comb))
                kwargs[name] = kw
                sets.append(
                    _ffi.make_params(kw["w"], kw.get("x", 0.0), kw.get("quantile", 0.5), kw["y_list"],
                                     kw["anc_allele_available"], n_src=n_eff)  # fmt: skip
                )
            for p in ploidy[: 2 + len(src_comb)]:
                _check_ploidy(p)
            n_sites = int(pos.size)
            res = four = dd = nsnps_all = None
            if names and windows and n_sites:
                pid = id(pos)
                if pid not in pos_dev_cache:
                    pos_dev_cache[pid] = torch.as_tensor(np.ascontiguousarray(pos, dtype=np.int32)).to(eng.device)
                pos_dev = pos_dev_cache[pid]
                lo, hi = eng.window_bounds(pos_dev, [w[0] for w in windows], [w[1] for w in windows])
                nsnps_all = (hi - lo).cpu().numpy()
                if uq_names:
                    counts = torch.stack(
                        [counts_rows[("ref", ref_pop)], counts_rows[("tgt", tgt_pop)]]
                        + [counts_rows[("src", s)] for s in src_comb[:n_eff]]
                    )
                    tgt_freq, flags, _ = eng.site_flags(counts, ploidy[: 2 + n_eff], sets)
                    res = eng.window_stats(tgt_freq, flags, sets, lo, hi, pos=pos_dev)
                if four_names:  # every source of the combination, with its own ploidy (fd_statistic.py:63-74)
                    if len(src_ploidies) < len(src_comb):
                        raise IndexError("list index out of range")
                    rows = [counts_rows[("ref", ref_pop)], counts_rows[("tgt", tgt_pop)]] + [
                        counts_rows[("src", s)] for s in src_comb
                    ]
                    pl4 = [ploidy[0], ploidy[1]] + list(src_ploidies[: len(src_comb)])
                    if out_pop is not None:
                        rows.append(counts_rows[("outgroup", out_pop)])
                        pl4.append(pc.get_ploidy("outgroup", out_pop))
                    for p in pl4:
                        _check_ploidy(p)
                    freqs = eng.site_freqs(torch.stack(rows), pl4)
                    four = eng.window_fourpop(freqs, len(src_comb), out_pop is not None, lo, hi).cpu().numpy()
                if want_dd:  # per source population: two streaming passes per pair of its individuals
                    dd = np.stack(
                        [
                            eng.window_dd(
                                eng.site_absdiff(tiled[("ref", ref_pop)], tiled[("src", s)]), tiled[("ref", ref_pop)].n_ind,
                                eng.site_absdiff(tiled[("tgt", tgt_pop)], tiled[("src", s)]), tiled[("tgt", tgt_pop)].n_ind,
                                lo, hi,
                            ).cpu().numpy()
                            for s in src_comb
                        ],
                        axis=1,
                    )  # fmt: skip
            for wi, (start, end) in enumerate(windows):
                nsnps = int(nsnps_all[wi]) if nsnps_all is not None else 0
                item = self._new_item(wg.chr_name, start, end, ref_pop, tgt_pop, src_comb, out_pop, nsnps)
                if nsnps == 0:  # window without sites: the reference's None-matrix branch
                    self._fill_missing(item, names)
                    items.append(item)
                    continue
                for name in four_names:  # one value per source, Python floats like the reference
                    item[name] = [float(v) for v in four[wi, :, _FOURPOP.index(name)]]
                if want_dd:  # np.float64 values, as np.mean returns them (dd_statistic.py:74)
                    item["DD"] = [np.float64(v) for v in dd[wi]]
                for si, name in enumerate(uq_names):
                    rec = res.records[si, wi]
                    if name == "U":
                        item["cdd_pos"][name] = res.u_list(si, wi).astype(pos.dtype, copy=True)
                        item[name] = int(rec["u_count"])
                    elif int(rec["n_cond"]) == 0:
                        item["cdd_pos"][name] = np.array([])
                        item[name] = np.nan
                    else:
                        item["cdd_pos"][name] = res.q_list(si, wi).astype(pos.dtype, copy=True)
                        item[name] = np.float64(rec["q"])
                items.append(item)
        return items

    # -- output ----------------------------------------------------------------------------

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
