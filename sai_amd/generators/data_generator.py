"""Generator protocol (mirror of sai/generators/data_generator.py:24-49)."""

from abc import ABC, abstractmethod


class DataGenerator(ABC):
    """``get()`` yields keyword dictionaries for ``DataPreprocessor.run``."""

    @abstractmethod
    def get(self, **kwargs):
        """Yield (or return an iterable of) parameter dictionaries."""
