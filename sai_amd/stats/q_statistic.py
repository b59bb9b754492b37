"""Q statistic (mirror of sai/stats/q_statistic.py:28-104) on the HIP path."""

from __future__ import annotations

from typing import Any, Dict

import numpy as np

from ..registries.stat_registry import STAT_REGISTRY
from ._window import run_single_window
from .generic_statistic import GenericStatistic


@STAT_REGISTRY.register("Q")
class QStatistic(GenericStatistic):
    """Quantile of the target frequency over the sites passing the ref/source conditions
    (Racimo et al. 2017), numpy 'linear' interpolation; NaN when no site qualifies."""

    STAT_NAME = "Q"

    def compute(self, **kwargs) -> Dict[str, Any]:
        required = ["pos", "w", "y_list", "anc_allele_available", "quantile"]
        if missing := [k for k in required if k not in kwargs]:  # q_statistic.py:70-72
            raise ValueError(f"Missing required argument(s): {', '.join(missing)}")
        pos = kwargs["pos"]
        rec, _, idx_q = run_single_window(
            self, kwargs["w"], None, kwargs["quantile"], kwargs["y_list"], kwargs["anc_allele_available"]
        )
        if len(pos) != int(rec["n_sites"]):
            # the reference's `pos[condition]` (q_statistic.py:93) with a boolean mask over the matrix rows:
            # numpy's IndexError when `pos` is not as long as the matrices
            raise IndexError(
                f"boolean index did not match indexed array along axis 0; size of axis is {len(pos)} "
                f"but size of corresponding boolean axis is {int(rec['n_sites'])}"
            )
        if int(rec["n_cond"]) == 0:  # q_statistic.py:96-98
            return {"name": self.STAT_NAME, "value": np.nan, "cdd_pos": np.array([])}
        # q_statistic.py:100-104
        return {"name": self.STAT_NAME, "value": np.float64(rec["q"]), "cdd_pos": pos[idx_q]}
