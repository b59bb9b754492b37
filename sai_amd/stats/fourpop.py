"""fd, df, Danc, Dplus (mirrors of sai/stats/{fd,df,danc,dplus}_statistic.py) on the HIP path.

All four are ratios of ABBA-BABA pattern sums over the sites of a window
(sai/stats/stat_utils.py:171-272); one GPU evaluation (site_counts -> site_freqs ->
window_fourpop) yields all of them for every source population, so the four classes share it.
"""

from __future__ import annotations

from typing import Any, Dict

import numpy as np

from ..registries.stat_registry import STAT_REGISTRY
from .generic_statistic import GenericStatistic
from .stat_utils import _check_ploidy

FOURPOP_ORDER = ("fd", "df", "Danc", "Dplus")


def evaluate_fourpop(stat: GenericStatistic) -> np.ndarray:
    """[n_src][4] = fd, df, Danc, Dplus of one window (the whole matrices)."""
    import torch

    from ..engine import Engine

    srcs = list(stat.src_gts_list)
    n_src = len(srcs)
    ploidy = [stat.ref_ploidy, stat.tgt_ploidy] + list(stat.src_ploidy_list)[:n_src]
    if len(ploidy) != 2 + n_src:
        raise IndexError("list index out of range")  # src_ploidy_list[i] in the reference
    mats = [stat.ref_gts, stat.tgt_gts] + srcs
    if stat.out_gts is not None:
        mats.append(stat.out_gts)
        ploidy.append(1 if stat.out_ploidy is None else stat.out_ploidy)
    for p in ploidy:
        _check_ploidy(p)
    eng = Engine.get()
    pops = [eng.tile(m) for m in mats]
    n_sites = pops[0].n_sites
    if any(p.n_sites != n_sites for p in pops):
        raise ValueError("genotype matrices must have the same number of sites")
    counts = eng.site_counts(pops)
    lo = torch.zeros(1, dtype=torch.int32, device=eng.device)
    hi = torch.full((1,), n_sites, dtype=torch.int32, device=eng.device)
    return eng.fourpop_windows(counts, ploidy, n_src, stat.out_gts is not None, lo, hi)[0].cpu().numpy()


class _FourPopStatistic(GenericStatistic):
    STAT_NAME = ""

    def compute(self, **kwargs) -> Dict[str, Any]:
        vals = evaluate_fourpop(self)[:, FOURPOP_ORDER.index(self.STAT_NAME)]
        return {"name": self.STAT_NAME, "value": [float(v) for v in vals]}


@STAT_REGISTRY.register("fd")
class FdStatistic(_FourPopStatistic):
    """Dynamic estimator of the proportion of introgression (Martin et al. 2015):
    (abba - baba) / (abba_d - baba_d) with tgt and src replaced by max(tgt, src) in the denominator."""

    STAT_NAME = "fd"


@STAT_REGISTRY.register("df")
class DfStatistic(_FourPopStatistic):
    """Distance fraction (Pfeifer & Kapan 2019): (abba - baba) / (abba + baba + 2 bbaa)."""

    STAT_NAME = "df"


@STAT_REGISTRY.register("Danc")
class DancStatistic(_FourPopStatistic):
    """D_anc (Lopez Fang et al. 2024): (baaa - abaa) / (baaa + abaa)."""

    STAT_NAME = "Danc"


@STAT_REGISTRY.register("Dplus")
class DplusStatistic(_FourPopStatistic):
    """D+ (Lopez Fang et al. 2024): (abba - baba + baaa - abaa) / (abba + baba + baaa + abaa)."""

    STAT_NAME = "Dplus"


@STAT_REGISTRY.register("DD")
class DdStatistic(GenericStatistic):
    """DD (mirror of sai/stats/dd_statistic.py:28-77): per source population, the mean over its
    individuals of (mean city-block distance to the reference individuals - mean city-block
    distance to the target individuals) over the window's sites, on the raw dosage values."""

    STAT_NAME = "DD"

    def compute(self, **kwargs) -> Dict[str, Any]:
        import torch

        from ..engine import Engine

        eng = Engine.get()
        ref, tgt = eng.tile(self.ref_gts), eng.tile(self.tgt_gts)
        n_sites = ref.n_sites
        lo = torch.zeros(1, dtype=torch.int32, device=eng.device)
        hi = torch.full((1,), n_sites, dtype=torch.int32, device=eng.device)
        values = []
        for src_gts in self.src_gts_list:
            src = eng.tile(src_gts)
            if src.n_sites != n_sites or tgt.n_sites != n_sites:
                raise ValueError("genotype matrices must have the same number of sites")
            dd = eng.window_dd(eng.site_absdiff(ref, src), ref.n_ind, eng.site_absdiff(tgt, src), tgt.n_ind, lo, hi)
            values.append(np.float64(dd[0].item()))
        return {"name": self.STAT_NAME, "value": values}
