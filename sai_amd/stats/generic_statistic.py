"""Statistic base class (mirror of sai/stats/generic_statistic.py:26-93)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict, Optional

import numpy as np


class GenericStatistic(ABC):
    """Holds the per-window genotype matrices ([sites][individuals], negative = missing) and
    ploidies exactly like the reference constructor (generic_statistic.py:34-76); nothing is
    copied or uploaded until ``compute`` is called."""

    def __init__(
        self,
        ref_gts: np.ndarray,
        tgt_gts: np.ndarray,
        ref_ploidy: int,
        tgt_ploidy: int,
        src_gts_list: list[np.ndarray],
        src_ploidy_list: list[int],
        out_gts: Optional[np.ndarray] = None,
        out_ploidy: Optional[int] = None,
    ):
        self.ref_gts = ref_gts
        self.tgt_gts = tgt_gts
        self.src_gts_list = src_gts_list
        self.out_gts = out_gts
        self.ref_ploidy = ref_ploidy
        self.tgt_ploidy = tgt_ploidy
        self.src_ploidy_list = src_ploidy_list
        self.out_ploidy = out_ploidy

    @abstractmethod
    def compute(self, **kwargs) -> Dict[str, Any]:
        """Return ``{"name", "value", "cdd_pos"}``."""
