"""Statistics of the U/Q path; importing this package registers "U" and "Q"
(mirror of sai/stats/__init__.py)."""

from .generic_statistic import GenericStatistic
from .q_statistic import QStatistic
from .stat_utils import calc_freq, compute_matching_loci
from .u_statistic import UStatistic

__all__ = ["GenericStatistic", "UStatistic", "QStatistic", "calc_freq", "compute_matching_loci"]
