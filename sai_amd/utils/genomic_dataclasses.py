"""Per-population chromosome block (mirror of sai/utils/genomic_dataclasses.py:25-46)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class ChromosomeData:
    """POS int32 [sites]; REF/ALT allele strings (or None for synthetic data); GT = unphased
    ALT dosage, int8 [sites][individuals], negative = missing (the reference holds the same
    matrix as int64 after utils.py:410)."""

    POS: np.ndarray
    REF: Optional[list]
    ALT: Optional[list]
    GT: np.ndarray
