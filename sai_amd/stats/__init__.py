"""Statistics on the HIP path; importing this package registers "U", "Q" and the ABBA-BABA
family "fd", "df", "Danc", "Dplus" (mirror of sai/stats/__init__.py; "DD" is not built)."""

from .fourpop import DancStatistic, DfStatistic, DplusStatistic, FdStatistic
from .generic_statistic import GenericStatistic
from .q_statistic import QStatistic
from .stat_utils import calc_freq, compute_matching_loci
from .u_statistic import UStatistic

__all__ = [
    "GenericStatistic",
    "UStatistic",
    "QStatistic",
    "FdStatistic",
    "DfStatistic",
    "DancStatistic",
    "DplusStatistic",
    "calc_freq",
    "compute_matching_loci",
]
