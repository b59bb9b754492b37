"""Statistics on the HIP path; importing this package registers "U", "Q" and the ABBA-BABA
family "fd", "df", "Danc", "Dplus" and "DD" (mirror of sai/stats/__init__.py)."""

from .fourpop import DancStatistic, DdStatistic, DfStatistic, DplusStatistic, FdStatistic
from .generic_statistic import GenericStatistic
from .q_statistic import QStatistic
from .stat_utils import calc_four_pops_freq, calc_freq, calc_pattern_sum, compute_matching_loci
from .u_statistic import UStatistic

__all__ = [
    "GenericStatistic",
    "UStatistic",
    "QStatistic",
    "FdStatistic",
    "DfStatistic",
    "DancStatistic",
    "DplusStatistic",
    "DdStatistic",
    "calc_freq",
    "calc_four_pops_freq",
    "calc_pattern_sum",
    "compute_matching_loci",
]
