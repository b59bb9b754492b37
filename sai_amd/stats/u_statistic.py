"""U statistic (mirror of sai/stats/u_statistic.py:28-99) on the HIP path."""

from __future__ import annotations

from typing import Any, Dict

from ..registries.stat_registry import STAT_REGISTRY
from ._window import run_single_window
from .generic_statistic import GenericStatistic


@STAT_REGISTRY.register("U")
class UStatistic(GenericStatistic):
    """Number of sites with ref_freq < w, tgt_freq > x and every source matching its condition
    (Racimo et al. 2017); ``compute`` keeps the reference's keyword contract."""

    STAT_NAME = "U"

    def compute(self, **kwargs) -> Dict[str, Any]:
        required = ["pos", "w", "x", "y_list", "anc_allele_available"]
        if missing := [k for k in required if k not in kwargs]:  # u_statistic.py:70-72
            raise ValueError(f"Missing required argument(s): {', '.join(missing)}")
        pos = kwargs["pos"]
        rec, idx_u, _ = run_single_window(
            self, kwargs["w"], kwargs["x"], None, kwargs["y_list"], kwargs["anc_allele_available"]
        )
        # u_statistic.py:94-99: positions of the counted sites, count as a Python int
        return {"name": self.STAT_NAME, "value": int(rec["u_count"]), "cdd_pos": pos[idx_u]}
