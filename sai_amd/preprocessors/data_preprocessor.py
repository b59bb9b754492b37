"""Preprocessor protocol (mirror of sai/preprocessors/data_preprocessor.py:25-94)."""

from abc import ABC, abstractmethod
from typing import Any


class DataPreprocessor(ABC):
    """``run(**params)`` computes items for one unit of work; ``process_items(items)`` writes
    them out.  Instances stay picklable (no GPU handles are stored on them)."""

    @abstractmethod
    def run(self, **kwargs) -> Any:
        """Process one work unit and return its items."""

    @abstractmethod
    def process_items(self, items: Any, **kwargs) -> None:
        """Persist the items returned by ``run``."""
