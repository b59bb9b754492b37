"""STAT_REGISTRY (mirror of sai/registries/stat_registry.py:30)."""

from .generic_registry import GenericRegistry


class StatRegistry(GenericRegistry):
    """Registry of statistic classes."""


STAT_REGISTRY = StatRegistry()
