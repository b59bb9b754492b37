from .global_config import GlobalConfig
from .ploidy_config import PloidyConfig
from .pop_config import PopConfig
from .stat_config import StatConfig

__all__ = ["GlobalConfig", "PloidyConfig", "PopConfig", "StatConfig"]
