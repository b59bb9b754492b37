from .generic_registry import GenericRegistry
from .stat_registry import STAT_REGISTRY, StatRegistry

__all__ = ["GenericRegistry", "StatRegistry", "STAT_REGISTRY"]
